import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cmax_slam_amd import _lib, evaluator, synth
p = synth.config2(1_000_000)
ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
pts = np.array([np.array(p.omega_true, float) * s for s in np.linspace(0, 1, 8)])
xs = np.vstack([pts[i % 8] for i in range(400)])
ev.eval_each(xs, True)
for mode in ("fence", "nofence", "sleep100us"):
    acc = np.zeros(20)
    for rep in range(30):
        ev.eval_each(xs[:64], True)
        if mode == "fence":
            torch.cuda.synchronize()
        elif mode == "sleep100us":
            t = time.perf_counter()
            while time.perf_counter() - t < 100e-6:
                pass
        ts = [time.perf_counter()]
        for i in range(20):
            ev.eval(pts[i % 8], True)
            ts.append(time.perf_counter())
        acc += np.diff(ts) * 1e6
    print(mode, "per-evaluation us (mean of 30):", " ".join("%.1f" % v for v in acc / 30))
