"""How a K = 20 timed region (the driver's bench command) depends on what the GPU did just before it: an idle gap (the calibration's
read-back), then N untimed evaluations, then 20 timed ones -- against the sustained rate of a 1000-step region.
python tools/k20_preroll.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402

p = synth.config2(1_000_000)
ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
ev.eval(np.zeros(3), True)
pts = np.array([np.array(p.omega_true, float) * s for s in np.linspace(0, 1, 8)])
xs = lambda n: np.vstack([pts[i % 8] for i in range(n)])  # noqa: E731
ev.eval_each(xs(2000), True)
t0 = time.perf_counter(); ev.eval_each(xs(1000), True); sustained = (time.perf_counter() - t0) / 1000 * 1e3
print("sustained (1000 steps): %.4f ms" % sustained)
for gap_ms in (0.0, 2.0, 20.0):
    for n in (0, 16, 64, 256, 1024):
        r = []
        for rep in range(7):
            ev.eval_each(xs(300), True)
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
            if n:
                ev.eval_each(xs(n), True)
            x20 = xs(20)
            t0 = time.perf_counter(); ev.eval_each(x20, True); r.append((time.perf_counter() - t0) / 20 * 1e3)
        r.sort()
        print("idle gap %5.1f ms, pre-roll %4d evaluations: K = 20 region %.4f ms per step (median of 7; min %.4f max %.4f)  x%.3f of sustained"
              % (gap_ms, n, r[3], r[0], r[-1], r[3] / sustained), flush=True)
