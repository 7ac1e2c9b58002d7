"""Does the kernel duration bench.py measures live (HIP events carried by the dispatch, cmx_timing_*) agree with rocprofv3's?  One
front-end context, BASELINE config 2, evaluations cycling 8 points of a cold-start solve's range -- bench.py's pattern -- in the loops the
bench uses.  Run plain for the live numbers; under `rocprofv3 --kernel-trace --stats` for the trace's (LIVE=0: no timed launches at all).
python tools/live_vs_trace.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402

live = os.environ.get("LIVE", "1") != "0"
p = synth.config2(1_000_000)
ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
ev.eval(np.zeros(3), True)
pts = np.array([np.array(p.omega_true, float) * s for s in np.linspace(0, 1, 8)])
xs = lambda n: np.vstack([pts[i % 8] for i in range(n)])  # noqa: E731
ev.eval_each(xs(3000), True)
t0 = time.perf_counter(); ev.eval_each(xs(2000), True); print("native loop, nothing timed: %.4f ms per evaluation" % ((time.perf_counter() - t0) / 2000 * 1e3))
if live:
    def show(name, tim):
        print("%-58s" % name, " ".join("%s=%.2f us (%d)" % (k, 1e3 * v[0] / v[1], v[1]) for k, v in tim.items() if v[1]), flush=True)
    for which, label in ((["splat"], "only splat timed"), (["gather"], "only gather timed"), (True, "every class timed")):
        ev.timing_enable(which); ev.timing_get()
        t0 = time.perf_counter(); ev.eval_each(xs(400), True); el = (time.perf_counter() - t0) / 400 * 1e3
        show("native loop, %s (%.4f ms / evaluation):" % (label, el), ev.timing_get())
        ev.timing_get()
        for i in range(200):
            ev.eval(pts[i % 8], True)
        show("python loop, %s:" % label, ev.timing_get())
        ev.timing_enable(False)
ev.close()
