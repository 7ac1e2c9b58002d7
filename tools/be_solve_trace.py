"""Back-end solves (config 3) in a loop, for `rocprofv3 --kernel-trace` + tools/gap_analysis.py: where a window's solve spends its
time.  Usage on the GPU box: python tools/be_solve_trace.py [solves]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = synth.config3(5_000_000)
ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
              w.sample_rate, w.sigma, _lib.VARIANCE)
ev.set_option(_lib.OPT_REUSE_IMAGE, 1)
ev.setupProblemAndOptimize()
best = 1e9
for _ in range(n):
    t0 = time.perf_counter()
    x, rep = ev.setupProblemAndOptimize()
    best = min(best, time.perf_counter() - t0)
print("best %.4f ms per solve; %s" % (best * 1e3, rep))
