"""-m gpu: the ABI's error contract -- every misuse returns a status code (never aborts the host, unlike the
reference's CHECK / assert / .at() paths) and leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def p():
    return synth.frontend_packet(5_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=61)


def test_call_sequence_errors(hip, p):
    L = _lib.lib()
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    c, g = C.c_double(), (C.c_double * 3)()
    om = (C.c_double * 3)(0.1, 0.2, 0.3)
    assert L.cmx_frontend_eval(fe._ctx, om, C.byref(c), g) == _lib.ERR_STATE            # before set_packet
    assert L.cmx_frontend_finish(fe._ctx, C.byref(c), g) == _lib.ERR_STATE              # finish without accumulate
    assert b"set_packet" in L.cmx_last_error(fe._ctx) or b"accumulate" in L.cmx_last_error(fe._ctx)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    assert L.cmx_frontend_accumulate(fe._ctx, om, 0) == _lib.OK
    assert L.cmx_frontend_finish(fe._ctx, C.byref(c), g) == _lib.ERR_STATE              # gradient asked, planes absent
    assert L.cmx_frontend_finish(fe._ctx, C.byref(c), None) == _lib.OK
    assert L.cmx_frontend_eval(fe._ctx, None, C.byref(c), g) == _lib.ERR_INVALID_ARG
    assert L.cmx_frontend_eval(fe._ctx, om, None, g) == _lib.ERR_INVALID_ARG
    assert L.cmx_backend_eval(fe._ctx, om, C.byref(c), g) == _lib.ERR_STATE              # wrong context kind
    assert L.cmx_frontend_finish_end(fe._ctx, C.byref(c), g) == _lib.ERR_STATE           # finish_end without begin
    assert L.cmx_set_option(fe._ctx, 99, 1) == _lib.ERR_INVALID_ARG
    assert L.cmx_set_option(fe._ctx, _lib.OPT_GRAD_MODE, 7) == _lib.ERR_INVALID_ARG
    # the context survives all of the above
    cc, gg = fe.eval((0.1, 0.2, 0.3))
    assert np.isfinite(cc) and np.all(np.isfinite(gg))


def test_bad_creation_arguments(hip, p):
    L = _lib.lib()
    ctx = _lib.ctx_p()
    lut = np.ascontiguousarray(p.lut).ctypes.data_as(_lib.c_dp)
    assert L.cmx_frontend_create(C.byref(ctx), 0, 0, 10, lut) == _lib.ERR_INVALID_ARG
    assert L.cmx_frontend_create(C.byref(ctx), 0, 10, 10, None) == _lib.ERR_INVALID_ARG
    assert L.cmx_frontend_create(C.byref(ctx), 99, p.W, p.H, lut) == _lib.ERR_INVALID_ARG   # no such device
    assert L.cmx_backend_create(C.byref(ctx), 0, p.W, p.H, lut, 2, 2) == _lib.ERR_INVALID_ARG
    L.cmx_destroy(None)  # harmless


def test_bad_packet_arguments(hip, p):
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    with pytest.raises(hip.CmaxHipError) as e:
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, event_batch_size=0)
    assert e.value.status == _lib.ERR_INVALID_ARG
    with pytest.raises(hip.CmaxHipError) as e:
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, blur_sigma=5.0)  # radius 20 > 12
    assert e.value.status == _lib.ERR_INVALID_ARG
    with pytest.raises(ValueError):
        fe.set_packet(p.x[:10], p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy)
    # a failed set_packet leaves no half-installed packet behind
    with pytest.raises(hip.CmaxHipError):
        fe.eval((0, 0, 0))


def test_backend_argument_errors(hip):
    w = synth.backend_window(3_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 256, 128, 2, 5, 1, 0.2, seed=62)
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    args = (w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns)
    for bad in (dict(num_fixed=9), dict(num_fixed=-1)):
        with pytest.raises(hip.CmaxHipError) as e:
            be.set_window(*args, bad["num_fixed"], w.t_next_win_beg_ns)
        assert e.value.status == _lib.ERR_INVALID_ARG
    with pytest.raises(hip.CmaxHipError):
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, 0, 1, w.t_next_win_beg_ns)  # dt_ns = 0
    with pytest.raises(hip.CmaxHipError):
        be.set_window(*args, 1, w.t_next_win_beg_ns, event_sample_rate=0)
    be.set_window(*args, 1, w.t_next_win_beg_ns)
    with pytest.raises(ValueError):
        be.eval(np.zeros(3))  # wrong parameter count is caught before crossing the ABI
    with pytest.raises(hip.CmaxHipError):
        be.get_plane(99)


def test_round5_entry_points_reject_bad_arguments(hip):
    """cmx_events_create_group / cmx_events_devices / cmx_comm_info / cmx_backend_set_window_from: argument errors come back as status
    codes with a message, nothing is half-installed."""
    import ctypes as C
    from cmax_slam_amd import _lib, synth
    L = _lib.lib()
    h = C.c_void_p()
    dv = (C.c_int * 2)(0, 99)
    assert L.cmx_events_create_group(C.byref(h), dv, 2, 64, 48, 1000) == _lib.ERR_INVALID_ARG and not h.value      # no such device
    assert L.cmx_events_create_group(C.byref(h), dv, 0, 64, 48, 1000) == _lib.ERR_INVALID_ARG and not h.value      # empty list
    assert L.cmx_events_create_group(C.byref(h), None, 1, 64, 48, 1000) == _lib.ERR_INVALID_ARG
    dv0 = (C.c_int * 3)(0, 0, 0)
    assert L.cmx_events_create_group(C.byref(h), dv0, 3, 64, 48, 1000) == _lib.OK and h.value
    out = (C.c_int * 4)()
    assert L.cmx_events_devices(h, out, 4) == 1 and out[0] == 0               # members sharing a device share its replica
    assert L.cmx_events_devices(None, out, 4) == 0
    L.cmx_events_destroy(h)

    w = synth.backend_window(5_000, 64, 48, 55.0, 57.0, 31.5, 23.5, 128, 64, 2, 5, 0, 0.2, seed=5)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    info = be.comm_info()
    assert info == {"rank": 0, "nranks": 1, "transport": "none"}
    assert L.cmx_comm_info(None, None, None, None) == _lib.ERR_INVALID_ARG
    k = np.ascontiguousarray(w.knots_init, np.float64)
    rc = L.cmx_backend_set_window_from(be._ctx, None, 0, 10, w.order, w.K, k.ctypes.data_as(_lib.c_dp), int(w.start_ns), int(w.dt_ns), 0,
                                       int(w.t_next_win_beg_ns), 100, 1, 1.0, 0, None)
    assert rc == _lib.ERR_INVALID_ARG and b"null event store" in L.cmx_last_error(be._ctx)
    store = hip.EventStore(w.W + 1, w.H, 10_000)                                  # another sensor
    store.push(w.x, w.y, w.t_ns)
    with pytest.raises(hip.CmaxHipError) as e:
        be.set_window_from(store, 0, len(w.x), w.order, w.knots_init, w.start_ns, w.dt_ns, 0, w.t_next_win_beg_ns)
    assert e.value.status == _lib.ERR_INVALID_ARG
    be.K, be.num_fixed = w.K, 0
    with pytest.raises(hip.CmaxHipError) as e:
        be.eval(np.zeros(w.P))
    assert e.value.status == _lib.ERR_STATE                                        # no window was installed
    store.close()
    be.close()
