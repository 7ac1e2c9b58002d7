import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cmax_slam_amd import _lib, evaluator, synth
p = synth.config2(1_000_000)
ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
x0 = np.array([0.3, -0.5, 0.2])
for m in (1, 5, 20, 100, 400):
    xs = np.tile(x0, (m, 1))
    for _ in range(5): ev.eval_each(xs, True)
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ev.eval_each(xs, True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts2 = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ev.eval_each(xs, True); ts2.append(time.perf_counter() - t0)
    print(m, "with fences: %.1f us total, %.2f us/step | no trailing fence: %.1f us" % (np.median(ts) * 1e6, np.median(ts) * 1e6 / m, np.median(ts2) * 1e6))
ts = []
for _ in range(50):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("idle synchronize: %.1f us" % (np.median(ts) * 1e6))
ev.timing_enable(["gather"], every=4)
xs = np.tile(x0, (20, 1)); ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev.eval_each(xs, True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("20 steps with live timing every 4: %.1f us" % (np.median(ts) * 1e6))
ev.timing_enable(False)
ev.timing_get()
rng = np.random.default_rng(0)
pts = x0 + rng.normal(0, 0.02, (8, 3))
for m in (20, 200, 1000):
    xs = np.vstack([pts[i % 8] for i in range(m)])
    for every in (0, 4, 16):
        if every: ev.timing_enable(["gather"], every=every)
        else: ev.timing_enable(False)
        ev.eval_each(xs, True); ev.timing_get()
        ts = []
        for _ in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter(); ev.eval_each(xs, True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            ev.timing_get()
        print("distinct points m=%d live timing every %d: %.2f us/step (min %.2f)" % (m, every, np.median(ts) * 1e6 / m, min(ts) * 1e6 / m))
