import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
import numpy as np
from cmax_slam_amd import _lib, evaluator, synth
p = synth.config1()
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
w = synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 1024, 512, 4, 8, 2, 0.25, seed=5)
be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
rng = np.random.default_rng(0)
t0 = time.time(); nfe = nbe = 0; costs_fe = []; costs_be = []
while time.time() - t0 < 45.0:
    for _ in range(50):
        x, rep = fe.setupProblemAndOptimize(rng.normal(0, 0.05, 3)); nfe += 1; costs_fe.append(rep["final_cost"])
    x, rep = be.setupProblemAndOptimize(); nbe += 1; costs_be.append(rep["final_cost"])
    # interleave plain evaluations and hinted sequences with wrong hints
    pt = rng.normal(0, 0.3, 3)
    fe.hint_next_df(float(rng.normal()), int(rng.integers(0, 5)))
    c0, _ = fe.eval(pt, False)
    if rng.random() < 0.5:
        c1, g1 = fe.eval(pt, True)
        assert abs(c1 - c0) <= 1e-6 * abs(c0)
    if nbe % 20 == 0:
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
cf, cb = np.array(costs_fe), np.array(costs_be)
print("soak ok: %d front-end solves, %d back-end solves in %.0f s; fe final cost %.6f .. %.6f, be %.6f .. %.6f; stats fe %s" % (nfe, nbe, time.time() - t0, cf.min(), cf.max(), cb.min(), cb.max(), {k: v for k, v in fe.stats().items() if "gated" in k or "spec" in k or "chain" in k}))
