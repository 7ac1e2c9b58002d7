"""-m gpu: events handed over as the reference holds them -- std::vector<dvs_msgs::Event>, an array of 16-byte records
{uint16 x, y; ros::Time ts {uint32 sec, nsec}; bool polarity} (src/frontend/ang_vel_estimator.cpp:68-147,
src/backend/pose_graph_optimizer.cpp:131-165) -- through the *_aos entry points: one packing pass straight from the records.
The device contents, hence every result, must be bit-identical to the SoA hand-over of the same events."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


def test_record_layout_is_dvs_msgs_event():
    assert _lib.DVS_EVENT_DTYPE.itemsize == 16
    ev = _lib.dvs_events([1, 2], [3, 4], [1_500_000_007, 2_000_000_000])
    lay = _lib.aos_layout_of(ev)
    assert (lay.stride, lay.off_x, lay.off_y, lay.off_sec, lay.off_nsec) == (16, 0, 2, 4, 8)
    assert list(ev["sec"]) == [1, 2] and list(ev["nsec"]) == [500_000_007, 0]


def test_frontend_packet_from_records_is_bit_identical(hip):
    p = synth.frontend_packet(80_021, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=61)
    ev = _lib.dvs_events(p.x, p.y, p.t_ns, polarity=np.arange(len(p.x)) & 1)
    a, b = hip.FrontendEvaluator(p.W, p.H, p.lut), hip.FrontendEvaluator(p.W, p.H, p.lut)
    for fe in (a, b):
        fe.set_deterministic(True)  # bitwise reproducible evaluations: any difference would be the hand-over's
    a.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    b.set_packet_aos(ev, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    for om in [(0.0, 0.0, 0.0), (0.6, -0.9, 0.4)]:
        ca, ga = a.eval(om)
        cb, gb = b.eval(om)
        assert ca == cb and np.array_equal(ga, gb), (om, ca, cb, ga, gb)
    # a wider record (a host struct with more fields behind the four): offsets are the layout's, not dvs_msgs'
    wide = np.zeros(len(p.x), np.dtype({"names": ["pad", "nsec", "sec", "y", "x"], "formats": ["<u8", "<u4", "<u4", "<u2", "<u2"],
                                        "offsets": [0, 8, 12, 16, 18], "itemsize": 24}))
    wide["x"], wide["y"], wide["sec"], wide["nsec"] = p.x, p.y, p.t_ns // 10**9, p.t_ns % 10**9
    b.set_packet_aos(wide, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    assert b.eval((0.6, -0.9, 0.4))[0] == a.eval((0.6, -0.9, 0.4))[0]


@pytest.mark.parametrize("rate", [1, 3])
def test_backend_window_from_records_is_bit_identical(hip, rate):
    w = synth.backend_window(60_001, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 2, 0.25, seed=62)
    ev = _lib.dvs_events(w.x, w.y, w.t_ns)
    a, b = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp), hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    grp = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0, 0])
    for be in (a, b):
        be.set_deterministic(True)
    a.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, rate, w.sigma)
    b.set_window_aos(ev, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, rate, w.sigma)
    grp.set_window_aos(ev, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, rate, w.sigma)
    rng = np.random.default_rng(4)
    for d in (np.zeros(w.P), rng.normal(0, 0.01, w.P)):
        ca, ga = a.eval(d, True)
        cb, gb = b.eval(d, True)
        cg, gg = grp.eval(d, True)
        assert ca == cb and np.array_equal(ga, gb)
        assert abs(cg - ca) < 1e-6 * abs(ca) and np.abs(gg - ga).max() < 1e-6 * np.abs(ga).max()  # (a group: sums in another order)
    assert np.array_equal(a.get_plane(_lib.PLANE_IL_OLD), b.get_plane(_lib.PLANE_IL_OLD))


def test_event_store_push_from_records(hip):
    p = synth.frontend_packet(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=63)
    ev = _lib.dvs_events(p.x, p.y, p.t_ns)
    s1, s2 = hip.EventStore(p.W, p.H, 100_000), hip.EventStore(p.W, p.H, 100_000)
    for k in range(0, len(p.x), 12_500):   # packets as ROS delivers them
        s1.push(p.x[k:k + 12_500], p.y[k:k + 12_500], p.t_ns[k:k + 12_500])
        s2.push_aos(ev[k:k + 12_500])
    a, b = hip.FrontendEvaluator(p.W, p.H, p.lut), hip.FrontendEvaluator(p.W, p.H, p.lut)
    for fe, st in ((a, s1), (b, s2)):
        fe.set_deterministic(True)
        fe.set_packet_from(st, 5_000, 40_000, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ca, ga = a.eval((0.5, -0.7, 0.3))
    cb, gb = b.eval((0.5, -0.7, 0.3))
    assert ca == cb and np.array_equal(ga, gb)


def test_aos_errors_match_the_soa_entry_points(hip):
    p = synth.frontend_packet(5_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=64)
    ev = _lib.dvs_events(p.x, p.y, p.t_ns)
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    bad = ev.copy()
    bad["x"][1234] = 240      # outside the 240-wide sensor
    with pytest.raises(hip.CmaxHipError) as e:
        fe.set_packet_aos(bad, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    assert e.value.status == _lib.ERR_EVENT_RANGE and "1234" in str(e.value)
    back = ev.copy()
    back["sec"][100:150] += 5   # batch [100, 200) now ends before it starts
    with pytest.raises(hip.CmaxHipError) as e:
        fe.set_packet_aos(back, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    assert e.value.status == _lib.ERR_TIME_ORDER
    lay = _lib.AosLayout(8, 0, 2, 4, 8)   # nsec beyond an 8-byte record
    L = _lib.lib()
    assert L.cmx_frontend_set_packet_aos(fe._ctx, 10, ev.ctypes.data, lay, 0, 1.0, 1.0, 0.0, 0.0, 100, 1.0, 0) == _lib.ERR_INVALID_ARG
    fe.set_packet_aos(ev[:0], p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)   # an empty packet is a packet
    assert fe.eval((0.1, 0.1, 0.1))[0] == 0.0
