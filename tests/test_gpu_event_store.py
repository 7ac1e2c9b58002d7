"""-m gpu: device-resident event store: packets and windows cut from an uploaded stream
must give exactly what cmx_frontend_set_packet / cmx_backend_set_window give on the same events, which in turn match
the oracle; overlapping packets need no re-upload; drop_before is deleteOldEvents."""
import time

import numpy as np
import pytest

from cmax_slam_amd import synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def test_overlapping_frontend_packets_from_the_store(hip, oracle):
    p = synth.frontend_packet(120_000, 240, 180, 200.0, 200.0, 119.5, 89.5, T=0.06, seed=51)
    store = hip.EventStore(p.W, p.H, capacity=200_000)
    # the stream arrives in chunks (eventsCallback, cmax_slam.cpp:147-161)
    for a in range(0, len(p.x), 25_000):
        store.push(p.x[a:a + 25_000], p.y[a:a + 25_000], p.t_ns[a:a + 25_000])
    assert (store.begin, store.end) == (0, len(p.x))
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    direct = hip.FrontendEvaluator(p.W, p.H, p.lut)
    direct.set_fast_path()
    om = (0.4, -0.6, 0.3)
    for first in (0, 20_000, 40_037, 70_000):  # heavily overlapping 50k-event packets
        n = 50_000
        sl = slice(first, first + n)
        t_ref = int(p.t_ns[first + n // 2])
        fe.set_packet_from(store, first, n, t_ref, p.fx, p.fy, p.cx, p.cy)
        direct.set_packet(p.x[sl], p.y[sl], p.t_ns[sl], t_ref, p.fx, p.fy, p.cx, p.cy)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy)
        ref.set_packet(p.x[sl], p.y[sl], p.t_ns[sl], t_ref)
        c_ref, g_ref = ref.eval(om)
        c, g = fe.eval(om)
        c_d, g_d = direct.eval(om)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
        # same events through the same kernels: equal up to the order of the per-chunk fp32 flushes
        assert rel_scalar(c, c_d) < 1e-8 and rel_vec(g, g_d) < 1e-6
    # deleteOldEvents: drop the first 60k, global indices keep their meaning
    store.drop_before(60_000)
    assert (store.begin, store.end) == (60_000, len(p.x))
    fe.set_packet_from(store, 70_000, 50_000, int(p.t_ns[95_000]), p.fx, p.fy, p.cx, p.cy)
    direct.set_packet(p.x[70_000:], p.y[70_000:], p.t_ns[70_000:], int(p.t_ns[95_000]), p.fx, p.fy, p.cx, p.cy)
    assert rel_scalar(fe.eval(om)[0], direct.eval(om)[0]) < 1e-8
    with pytest.raises(hip.CmaxHipError):  # dropped events are gone
        fe.set_packet_from(store, 10_000, 1000, int(p.t_ns[10_500]), p.fx, p.fy, p.cx, p.cy)
    with pytest.raises(hip.CmaxHipError):  # capacity is enforced
        store.push(p.x, p.y, p.t_ns + 10**9)
        store.push(p.x, p.y, p.t_ns + 2 * 10**9)


@pytest.mark.parametrize("rate", [1, 3])
def test_backend_window_from_the_store(hip, oracle, rate):
    w = synth.backend_window(60_001, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=52)
    store = hip.EventStore(w.W, w.H, capacity=100_000)
    store.push(w.x, w.y, w.t_ns)
    first, n = 5_000, 50_001
    sl = slice(first, first + n)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window_from(store, first, n, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                       100, rate)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, 100, rate)
    ref.set_window(w.x[sl], w.y[sl], w.t_ns[sl], w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = np.random.default_rng(1).normal(0, 0.005, w.P)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


def test_store_window_batch_times_are_validated_on_the_device(hip):
    """Windows cut from the store get their per-batch pose times (ros::Duration arithmetic) from a kernel; a batch time
    outside the spline's support must come back as CMX_ERR_SPLINE_RANGE exactly as on the host path, and the context
    must stay usable."""
    from cmax_slam_amd import _lib
    w = synth.backend_window(20_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 1, 0.2, seed=53)
    store = hip.EventStore(w.W, w.H, capacity=len(w.x))
    store.push(w.x, w.y, w.t_ns)
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    for kw in (dict(start_ns=w.start_ns + 10_000_000), dict(knots=w.knots_init[:3])):  # starts too late / ends too early
        with pytest.raises(hip.CmaxHipError) as e:
            be.set_window_from(store, 0, len(w.x), w.order, kw.get("knots", w.knots_init), kw.get("start_ns", w.start_ns),
                               w.dt_ns, 0, w.t_next_win_beg_ns)
        assert e.value.status == _lib.ERR_SPLINE_RANGE
        x = np.zeros(64)
        c = _lib.C.c_double()
        assert _lib.lib().cmx_backend_eval(be._ctx, x.ctypes.data_as(_lib.c_dp), _lib.C.byref(c), None) == _lib.ERR_STATE  # no half-installed window
    be.set_window_from(store, 0, len(w.x), w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    host = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    host.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = np.full(w.P, 0.004)
    (c0, g0), (c1, g1) = be.eval(d), host.eval(d)
    assert rel_scalar(c0, c1) < 1e-7 and rel_vec(g0, g1) < 1e-6  # same batch times, same events


def test_store_setup_is_cheaper_than_reupload(hip):
    p = synth.config2()
    store = hip.EventStore(p.W, p.H, capacity=len(p.x))
    store.push(p.x, p.y, p.t_ns)
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)           # warm allocations
    fe.set_packet_from(store, 0, len(p.x), p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    t0 = time.perf_counter()
    for _ in range(5):
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    t_host = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        fe.set_packet_from(store, 0, len(p.x), p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    t_store = (time.perf_counter() - t0) / 5
    print("set_packet %.3f ms, set_packet_from %.3f ms" % (t_host * 1e3, t_store * 1e3))
    assert t_store < t_host
