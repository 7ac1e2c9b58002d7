"""Per-kernel times of the back end at the reference's launch-file shapes (linear spline, K = 5, window 0.2 s; bench.py
launch_default_shapes): where an evaluation of a 200k-event window goes.   python tools/launch_default_kernels.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth
for name, (W, H, f, Wp, Hp, stride) in (("ijrr", (240, 180, 200.0, 1024, 512, 0.1)), ("ecrot_handheld", (1280, 720, 1000.0, 4096, 2048, 0.2))):
    for n_ev in (50_000, 200_000, 1_000_000):
        w = synth.backend_window(n_ev, W, H, f, f, (W - 1) / 2.0, (H - 1) / 2.0, Wp, Hp, 2, 5, 1, 0.2, dt_knots=0.05, seed=synth.SEED0 + 41, win_stride=stride)
        ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
        ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
        rng = np.random.default_rng(5)
        pts = np.vstack([rng.normal(0, 0.003, w.P) * s for s in (0.0, 0.3, 0.6, 1.0)] * 50)
        ev.eval_each(pts[:40], True)
        t0 = time.perf_counter(); ev.eval_each(pts, True); fdf = (time.perf_counter() - t0) / len(pts) * 1e6
        t0 = time.perf_counter(); ev.eval_each(pts, False); fc = (time.perf_counter() - t0) / len(pts) * 1e6
        ev.timing_enable(True); ev.timing_get()
        ev.eval_each(pts[:80], True); tim = ev.timing_get()
        ev.eval_each(pts[:80], False); timf = ev.timing_get()
        ev.timing_enable(False)
        ks = " ".join("%s=%.1f" % (k, 1e3 * v[0] / v[1]) for k, v in tim.items() if v[1])
        kf = " ".join("%s=%.1f" % (k, 1e3 * v[0] / v[1]) for k, v in timf.items() if v[1])
        print("%-15s %8d events  %dx%d  fdf %.1f us (kernels: %s)   f %.1f us (kernels: %s)" % (name, n_ev, Wp, Hp, fdf, ks, fc, kf), flush=True)
        ev.close()
