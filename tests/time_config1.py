"""BASELINE config 1 (ecrot_synth front end: 100k events, 240x180): the CPU oracle and the HIP path side by side --
one fdf evaluation and one full FR-CG solve from omega = 0 (the same restated driver over both)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np
from cmax_slam_amd import _lib, synth, evaluator, solver
from oracle import pyoracle as po

_lib.lib()  # load the library (and torch's HIP runtime) before anything is timed
p = synth.config1()
ref = po.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, po.VARIANCE)
ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
x0 = np.array([0.3, -0.5, 0.2])
ref.eval(x0)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    ref.eval(x0); n += 1
cpu_eval = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
def neg(v, want):
    c, g = ref.eval(v, want)
    return -c, (None if g is None else -g)
xs, rep = solver.frcg_minimize(neg, np.zeros(3), **solver.FRONTEND)
cpu_solve = time.perf_counter() - t0
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
for _ in range(20): fe.eval(x0)
fe.set_option(3, 0)
t0 = time.perf_counter()
for _ in range(300): c, g = fe.eval(x0)
gpu_eval = (time.perf_counter() - t0) / 300
fe.set_option(3, 1)
fe.setupProblemAndOptimize(np.zeros(3))
t0 = time.perf_counter()
for _ in range(20): xg, rg = fe.setupProblemAndOptimize(np.zeros(3))
gpu_solve = (time.perf_counter() - t0) / 20
cr, gr = ref.eval(x0)
print("config 1: CPU oracle fdf %.2f ms (%.2e ev/s), solve %.1f ms (%d iterations, %d+%d evaluations) -> omega %s"
      % (cpu_eval * 1e3, len(p.x) / cpu_eval, cpu_solve * 1e3, rep["iterations"], rep["n_f"], rep["n_df"], np.round(xs, 4)))
print("config 1: MI355X     fdf %.3f ms (%.2e ev/s), solve %.2f ms (%d iterations, %d+%d evaluations) -> omega %s"
      % (gpu_eval * 1e3, len(p.x) / gpu_eval, gpu_solve * 1e3, rg["iterations"], rg["n_f"], rg["n_df"], np.round(xg, 4)))
print("agreement at x0: contrast rel %.1e, gradient rel %.1e" % (abs(c - cr) / abs(cr), np.abs(g - gr).max() / np.abs(gr).max()))
