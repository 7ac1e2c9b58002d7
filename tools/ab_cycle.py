"""A/B of CMX_OPT_FUSED_IMAGE on bench.py's own pattern: the tile sort taken at omega = 0, then evaluations cycling through points of a
cold-start solve's range (0 .. the packet's true rate).  python tools/ab_cycle.py [npoints]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = synth.config2(1_000_000)
for fused in [int(v) for v in os.environ.get('AB_FUSED', '0,1,2,3,0,1,2,3').split(',')]:
    ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
    ev.set_option(_lib.OPT_FUSED_IMAGE, fused)
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    ev.eval(np.zeros(3), True)
    pts = np.array([np.array(p.omega_true, float) * s for s in np.linspace(0, 1, npts)])
    xs = np.vstack([pts[i % npts] for i in range(400)])
    ev.eval_each(xs, True)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ev.eval_each(xs, True)
        best = min(best, (time.perf_counter() - t0) / len(xs))
    ev.timing_enable(True)
    ev.timing_get()
    for i in range(64):
        ev.eval(pts[i % npts], True)
    tim = ev.timing_get()
    ev.timing_enable(False)
    st = ev.stats()
    print("fused=%d: %.4f ms per fdf cycling %d points; kernels(us): %s; rebins %d fallback %.5f fused %d redos %d"
          % (fused, best * 1e3, npts, " ".join("%s=%.1f" % (k, 1e3 * v[0] / v[1]) for k, v in tim.items() if v[1]), st["rebins"],
             st["fallback_frac"], st["fused_evals"], st["fused_redos"]) + " one-launch %d self-service %d timeouts %d" % (st["one_launch_evals"], st["self_serve_evals"], st["fused_timeouts"]), flush=True)
    ev.close()
