"""First-contact script for the GPU box: quick timing + parity printout of every mode (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmax_slam_amd import synth, evaluator
from oracle import pyoracle as po

p = synth.config2()
fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
ref = po.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
om = (0.3, -0.5, 0.2)
c, g = fe.eval(om)
t = time.time(); cr, gr = ref.eval(om); t_cpu = time.time() - t
print("fe gpu", c, g); print("fe cpu", cr, gr, "cpu s", t_cpu)
fe.timing_enable(True)
for want in (True, False):
    for _ in range(3): fe.eval(om, want)
    fe.timing_get()
    t = time.time(); n = 20
    for _ in range(n): fe.eval(om, want)
    dt = (time.time() - t) / n
    print("fe want_grad", want, "ms/eval", dt * 1e3, "Mev/s", len(p.x) / dt / 1e6, fe.timing_get())

fe.set_grad_mode(1)
c, g = fe.eval(om); print("fe adjoint", c, g, "rel", abs(c-cr)/cr, np.abs(g-gr).max()/np.abs(gr).max())
for _ in range(3): fe.eval(om, True)
fe.timing_get()
t = time.time(); n = 20
for _ in range(n): fe.eval(om, True)
dt = (time.time() - t) / n
print("fe ADJOINT fdf ms/eval", dt * 1e3, "Mev/s", len(p.x) / dt / 1e6, fe.timing_get())

fe.set_splat_mode(1)
c, g = fe.eval(om); print("fe adjoint+lds", c, g, "rel", abs(c-cr)/cr, np.abs(g-gr).max()/np.abs(gr).max(), fe.stats())
for want in (True, False):
    for _ in range(3): fe.eval(om, want)
    fe.timing_get()
    t = time.time(); n = 20
    for _ in range(n): fe.eval(om, want)
    dt = (time.time() - t) / n
    print("fe ADJOINT+LDS want_grad", want, "ms/eval", dt * 1e3, "Mev/s", len(p.x) / dt / 1e6, fe.timing_get(), fe.stats())
fe.timing_enable(False)
for want in (True, False):
    t = time.time(); n = 50
    for _ in range(n): fe.eval(om, want)
    dt = (time.time() - t) / n
    print("fe ADJOINT+LDS (no event timing) want_grad", want, "ms/eval", dt * 1e3, "Mev/s", len(p.x) / dt / 1e6)

w = synth.config3(1_000_000)
be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
rb = po.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
rb.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
d = np.zeros(w.P)
c, g = be.eval(d)
t = time.time(); cr, gr = rb.eval(d); t_cpu = time.time() - t
print("be gpu", c, g[:4]); print("be cpu", cr, gr[:4], "cpu s", t_cpu)
print("be rel", abs(c - cr) / cr, np.abs(g - gr).max() / np.abs(gr).max())
be.timing_enable(True)
for want in (True, False):
    for _ in range(2): be.eval(d, want)
    be.timing_get()
    t = time.time(); n = 10
    for _ in range(n): be.eval(d, want)
    dt = (time.time() - t) / n
    print("be want_grad", want, "ms/eval", dt * 1e3, "Mev/s", len(w.x) / dt / 1e6, be.timing_get())

be.set_grad_mode(1)
c, g = be.eval(d); print("be adjoint rel", abs(c - cr) / cr, np.abs(g - gr).max() / np.abs(gr).max())
for _ in range(2): be.eval(d, True)
be.timing_get()
t = time.time(); n = 10
for _ in range(n): be.eval(d, True)
dt = (time.time() - t) / n
print("be ADJOINT fdf ms/eval", dt * 1e3, "Mev/s", len(w.x) / dt / 1e6, be.timing_get())

be.set_splat_mode(1)
c, g = be.eval(d); print("be adjoint+lds rel", abs(c - cr) / cr, np.abs(g - gr).max() / np.abs(gr).max(), be.stats())
for want in (True, False):
    for _ in range(2): be.eval(d, want)
    be.timing_get()
    t = time.time(); n = 10
    for _ in range(n): be.eval(d, want)
    dt = (time.time() - t) / n
    print("be ADJOINT+LDS want_grad", want, "ms/eval", dt * 1e3, "Mev/s", len(w.x) / dt / 1e6, be.timing_get(), be.stats())
