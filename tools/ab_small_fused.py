"""CMX_OPT_FUSED_IMAGE forms on SMALL packets (the reference's own operating point: tens of thousands of events on a DAVIS sensor),
where an evaluation is launch latency more than work.  python tools/ab_small_fused.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402

shapes = [(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5), (100_000, 240, 180, 200.0, 200.0, 119.5, 89.5), (100_000, 320, 256, 280.0, 280.0, 159.5, 127.5),
          (30_000, 346, 260, 300.0, 300.0, 172.5, 129.5), (250_000, 640, 480, 588.1, 594.0, 339.8, 242.4)]
for (n, W, H, fx, fy, cx, cy) in shapes:
    p = synth.frontend_packet(n, W, H, fx, fy, cx, cy, seed=5)
    res = []
    for fused in (0, 1, 3, 0, 1, 3):
        fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
        fe.set_option(_lib.OPT_FUSED_IMAGE, fused)
        fe.set_option(_lib.OPT_REUSE_IMAGE, 0)
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        om = np.array(p.omega_true, float)
        fe.eval(om, True)
        rng = np.random.default_rng(1)
        xs = np.vstack([om + rng.normal(0, 0.01, 3) for _ in range(400)])
        fe.eval_each(xs, True)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); fe.eval_each(xs, True); best = min(best, (time.perf_counter() - t0) / len(xs))
        st = fe.stats()
        res.append("%d: %.2f us (self-service %d, timeouts %d, chunks %d)" % (fused, best * 1e6, st["self_serve_evals"], st["fused_timeouts"], st["chunks"]))
        fe.close()
    print("%7d events %dx%d | " % (n, W, H) + " | ".join(res), flush=True)
