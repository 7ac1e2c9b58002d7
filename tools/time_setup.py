import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmax_slam_amd import synth, evaluator
p = synth.config2()
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_fast_path()
for _ in range(3):
    t = time.perf_counter(); fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0); t1 = time.perf_counter()
    c, g = fe.eval((0.3, -0.5, 0.2)); t2 = time.perf_counter()      # first eval: includes the binning
    c, g = fe.eval((0.31, -0.5, 0.2)); t3 = time.perf_counter()
    x, rep = fe.setupProblemAndOptimize(np.zeros(3)); t4 = time.perf_counter()
    print("fe set_packet %.3f ms, first eval (with binning) %.3f ms, next eval %.3f ms, solve %.3f ms (%d evals)" % ((t1-t)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, rep["n_f"]+rep["n_df"]))
w = synth.config3()
be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
be.set_fast_path()
for _ in range(2):
    t = time.perf_counter(); be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns); t1 = time.perf_counter()
    c, g = be.eval(np.zeros(w.P)); t2 = time.perf_counter()
    c, g = be.eval(np.full(w.P, 1e-3)); t3 = time.perf_counter()
    print("be set_window %.3f ms, first eval (with binning) %.3f ms, next eval %.3f ms" % ((t1-t)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
# small packets cut from a device-resident event store (the end-to-end example's regime: 60k events at 240x180)
s = synth.event_stream(2e6, 0.2, 240, 180, 200.0, 200.0, 119.5, 89.5)
store = evaluator.EventStore(s.W, s.H, len(s.x)); store.push(s.x, s.y, s.t_ns)
fe2 = evaluator.FrontendEvaluator(s.W, s.H, s.lut); fe2.set_fast_path()
for k in range(4):
    first = 50_000 + 20_000 * k
    t = time.perf_counter(); fe2.set_packet_from(store, first, 60_000, int(s.t_ns[first + 30_000]), s.fx, s.fy, s.cx, s.cy); t1 = time.perf_counter()
    c, g = fe2.eval((0.2, 1.5, 0.3)); t2 = time.perf_counter()
    c, g = fe2.eval((0.21, 1.5, 0.3)); t3 = time.perf_counter()
    c = fe2.eval((0.22, 1.5, 0.3), False)[0]; t4 = time.perf_counter()
    print("60k-event packet from the store: set %.3f ms, first eval (with binning) %.3f ms, next fdf %.3f ms, f %.3f ms" % ((t1-t)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
