import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmax_slam_amd import synth, evaluator
p = synth.config2()
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_fast_path()
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
om = (0.3, -0.5, 0.2)
fe.set_option(3, 0)
for _ in range(5): fe.eval(om, True)
fe.timing_enable(True); fe.timing_get()
for _ in range(30): fe.eval(om, True)
t = fe.timing_get()
print("kernel us:", {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in t.items()}, fe.stats())
