"""-m gpu: cooperative scheduling between the two paths on one GPU (cmx_set_sched_class).  The reference's front end (a short
solve per packet, src/node.cpp:22) runs beside the back-end thread's window solves (src/cmax_slam.cpp:92); a BACKGROUND context
holds its next evaluation while an URGENT context of the same device is busy.  Host-side only: every result must be what the
contexts give alone; a background context must neither starve nor be delayed when no urgent context is active."""
import threading
import time

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _contexts(hip):
    p = synth.frontend_packet(200_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=181)
    w = synth.backend_window(200_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 1, 0.2, seed=182)
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    return p, w, fe, be


def test_classes_change_no_result_and_nobody_starves(hip, oracle):
    p, w, fe, be = _contexts(hip)
    rf = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    rf.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    rb = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    rb.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    om, d = np.array([0.3, -0.5, 0.2]), np.full(w.P, 0.003)
    (cf, gf), (cb, gb) = rf.eval(om), rb.eval(d)
    x_solo, rep_solo = fe.setupProblemAndOptimize(np.zeros(3))
    fe.set_sched_class(_lib.SCHED_URGENT)
    be.set_sched_class(_lib.SCHED_BACKGROUND)
    errors, done = [], {"fe": 0, "be": 0}
    stop = time.perf_counter() + 1.0

    def front():  # 100 % duty: evaluations and solves back to back -- the worst case for the background context
        try:
            while time.perf_counter() < stop:
                c, g = fe.eval(om)
                assert rel_scalar(c, cf) < RTOL and rel_vec(g, gf) < RTOL
                x, rep = fe.setupProblemAndOptimize(np.zeros(3))
                assert np.abs(x - x_solo).max() < 0.02 and abs(rep["iterations"] - rep_solo["iterations"]) <= 2
                done["fe"] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("front", repr(e)))

    def back():
        try:
            while time.perf_counter() < stop:
                c, g = be.eval(d)
                assert rel_scalar(c, cb) < RTOL and rel_vec(g, gb) < RTOL
                done["be"] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("back", repr(e)))

    ts = [threading.Thread(target=front), threading.Thread(target=back)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts)
    # the background context advanced although the urgent one never paused (its wait is capped at 5 ms in a row)
    assert done["fe"] >= 20 and done["be"] >= 20, done
    fe.close()
    be.close()


def test_background_alone_is_not_delayed(hip):
    """No urgent context active (one exists, idle for longer than the 20 us linger): a background evaluation starts at once."""
    p, w, fe, be = _contexts(hip)
    d = np.full(w.P, 0.003)
    for _ in range(20):
        be.eval(d)
    t0 = time.perf_counter()
    for _ in range(50):
        be.eval(d)
    normal = (time.perf_counter() - t0) / 50
    fe.set_sched_class(_lib.SCHED_URGENT)
    fe.eval(np.array([0.3, -0.5, 0.2]))     # an urgent call has happened on this device ...
    time.sleep(0.001)                        # ... and is long over
    be.set_sched_class(_lib.SCHED_BACKGROUND)
    t0 = time.perf_counter()
    for _ in range(50):
        be.eval(d)
    background = (time.perf_counter() - t0) / 50
    assert background < 1.5 * normal + 20e-6, (normal, background)
    with pytest.raises(Exception):
        be.set_sched_class(5)
    fe.close()
    be.close()


def test_background_group_beside_an_urgent_front_end(hip):
    """The one-process multi-GPU handle takes a class too (every member, their worker threads included, yields): a background
    group of two members evaluates beside an urgent front end and returns what the single context returns."""
    p, w, fe, be = _contexts(hip)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0])
    grp.set_fast_path()
    grp.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = np.full(w.P, 0.003)
    c0, g0 = be.eval(d)
    fe.set_sched_class(_lib.SCHED_URGENT)
    grp.set_sched_class(_lib.SCHED_BACKGROUND)
    om = np.array([0.3, -0.5, 0.2])
    errors, n = [], [0]
    stop = time.perf_counter() + 0.5

    def front():
        try:
            while time.perf_counter() < stop:
                fe.setupProblemAndOptimize(np.zeros(3))
                fe.eval(om)
                time.sleep(0.002)      # a front end with pauses: the group advances in them
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=front)
    t.start()
    while time.perf_counter() < stop:
        c, g = grp.eval(d)
        assert rel_scalar(c, c0) < 1e-6 and rel_vec(g, g0) < 1e-6
        n[0] += 1
    t.join(timeout=60)
    assert not errors and not t.is_alive() and n[0] >= 10, (errors, n)
    for e in (fe, be, grp):
        e.close()
