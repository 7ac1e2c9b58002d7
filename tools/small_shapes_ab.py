import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth
out = []
for n_ev in (50_000, 200_000):
    w = synth.backend_window(n_ev, 240, 180, 200.0, 200.0, 119.5, 89.5, 1024, 512, 2, 5, 1, 0.2, dt_knots=0.05, seed=synth.SEED0 + 41, win_stride=0.1)
    ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    rng = np.random.default_rng(5)
    pts = np.vstack([rng.normal(0, 0.003, w.P) * s for s in (0.0, 0.3, 0.6, 1.0)] * 100)
    ev.eval_each(pts[:40], True)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ev.eval_each(pts, True); best = min(best, (time.perf_counter() - t0) / len(pts) * 1e6)
    ev.set_option(_lib.OPT_REUSE_IMAGE, 1)
    ev.setupProblemAndOptimize()
    t0 = time.perf_counter()
    for _ in range(10): ev.setupProblemAndOptimize()
    out.append("be %dk fdf %.2f us solve %.1f us" % (n_ev // 1000, best, (time.perf_counter() - t0) / 10 * 1e6))
    ev.close()
p = synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=5)
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
for _ in range(5): fe.setupProblemAndOptimize(np.zeros(3))
t0 = time.perf_counter()
for _ in range(40): fe.setupProblemAndOptimize(np.zeros(3))
out.append("fe 60k solve %.1f us" % ((time.perf_counter() - t0) / 40 * 1e6))
print(" | ".join(out))
