"""-m gpu: the reference runs the front end on the ROS spin thread and the back end on its own thread, concurrently
(cmax_slam.cpp:92, node.cpp:22).  Two contexts driven from two host threads at the same time -- hand-overs (which
share the host packing pool), evaluations (each spinning on its own completion ticket) and C++ solves -- must give
exactly what they give alone."""
import threading

import numpy as np
import pytest

from cmax_slam_amd import synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def test_frontend_and_backend_threads_run_concurrently(hip, oracle):
    p = synth.frontend_packet(300_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=81)   # large enough to use the pool
    w = synth.backend_window(300_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 1, 0.2, seed=82)
    rf = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    rf.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    rb = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    rb.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    om, d = np.array([0.3, -0.5, 0.2]), np.full(w.P, 0.003)
    fe_ref, be_ref = rf.eval(om), rb.eval(d)
    errors, results = [], {}
    start = threading.Barrier(2)

    def front():
        try:
            fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
            fe.set_fast_path()
            start.wait()
            out = []
            for k in range(12):
                fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
                out.append(fe.eval(om))
                out.append((fe.eval(om, False)[0], None))
            x, rep = fe.setupProblemAndOptimize(np.zeros(3))
            results["fe"] = (out, rep)
        except Exception as e:  # noqa: BLE001
            errors.append(("front", repr(e)))

    def back():
        try:
            be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
            be.set_fast_path()
            start.wait()
            out = []
            for k in range(8):
                be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
                out.append(be.eval(d))
                out.append((be.eval(d, False)[0], None))
            x, rep = be.setupProblemAndOptimize()
            be.updateIG(200)
            results["be"] = (out, rep)
        except Exception as e:  # noqa: BLE001
            errors.append(("back", repr(e)))

    ts = [threading.Thread(target=front), threading.Thread(target=back)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts)
    for key, (c_ref, g_ref) in (("fe", fe_ref), ("be", be_ref)):
        out, rep = results[key]
        for c, g in out:
            assert rel_scalar(c, c_ref) < RTOL
            if g is not None:
                assert rel_vec(g, g_ref) < RTOL
        assert rep["final_cost"] < rep["initial_cost"] and rep["iterations"] >= 1
