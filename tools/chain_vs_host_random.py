"""Device-driven vs host-driven front-end solves from 2000 random starts (config 1), with two host-driven contexts beside each
other as the yardstick: the evaluations differ run to run by the order of the fp32 atomics, and FR-CG's stopping rules amplify that.
Usage on the GPU box: python tools/chain_vs_host_random.py"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
import numpy as np
from cmax_slam_amd import _lib, evaluator, synth
p = synth.config1()
def mk(chain):
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_option(_lib.OPT_CHAIN_SOLVE, chain)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    return fe
a, b, c = mk(1), mk(0), mk(0)
rng = np.random.default_rng(1)
bad = 0; d = []; dh = []
for i in range(2000):
    x0 = rng.normal(0, 0.05, 3)
    xa, ra = a.setupProblemAndOptimize(x0)
    xb, rb = b.setupProblemAndOptimize(x0)
    rel = abs(ra["final_cost"] - rb["final_cost"]) / abs(rb["final_cost"])
    d.append(rel)
    xc, rc = c.setupProblemAndOptimize(x0)
    dh.append(abs(rc["final_cost"] - rb["final_cost"]) / abs(rb["final_cost"]))
    if rel > 2e-3 or ra["initial_cost"] != rb["initial_cost"] and abs(ra["initial_cost"]-rb["initial_cost"]) > 1e-6*abs(rb["initial_cost"]):
        bad += 1
        if bad < 3: print(i, x0, ra, rb)
d = np.array(d); dh = np.array(dh)
print("host-driven vs host-driven (two contexts, same starts): median %.2e p99 %.2e max %.2e; > 2e-3: %d" % (np.median(dh), np.quantile(dh, 0.99), dh.max(), int((dh > 2e-3).sum())))
print("2000 random starts: chain vs host-driven final cost rel diff median %.2e p99 %.2e max %.2e; > 2e-3: %d; stats %s" % (np.median(d), np.quantile(d, 0.99), d.max(), bad, {k: v for k, v in a.stats().items() if "chain" in k}))
