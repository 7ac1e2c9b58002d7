"""-m gpu: set_packet queues its uploads on the context's stream and returns without waiting for them -- the caller's arrays
must nevertheless be free to change the moment the call returns, packets set back to back must not mix, and the tables kept
from the previous packet (same sigma) must be rebuilt when sigma changes."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def test_caller_buffers_are_free_after_set_packet_and_packets_do_not_mix(hip, oracle):
    W, H = 320, 240
    f = 0.9 * W
    packets = [synth.frontend_packet(200_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, seed=s) for s in (1, 2, 3)]
    fe = hip.FrontendEvaluator(W, H, packets[0].lut)
    fe.set_fast_path()
    om = [0.3, -0.5, 0.2]
    refs = []
    for p in packets:
        r = oracle.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        r.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        refs.append(r.eval(om))
    for rep in range(3):
        for p, (c_ref, g_ref) in zip(packets, refs):
            x, y, t = p.x.copy(), p.y.copy(), p.t_ns.copy()
            fe.set_packet(x, y, t, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
            x[:] = 0; y[:] = 0; t[:] = t[0]          # the caller reuses its arrays at once
            del x, y, t
            c, g = fe.eval(om)
            assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, rep
        # two packets set back to back, only the second evaluated
        a, b = packets[rep % 3], packets[(rep + 1) % 3]
        fe.set_packet(a.x, a.y, a.t_ns, a.t_ref_ns, a.fx, a.fy, a.cx, a.cy, a.batch, a.sigma, _lib.VARIANCE)
        fe.set_packet(b.x, b.y, b.t_ns, b.t_ref_ns, b.fx, b.fy, b.cx, b.cy, b.batch, b.sigma, _lib.VARIANCE)
        c, g = fe.eval(om)
        c_ref, g_ref = refs[(rep + 1) % 3]
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


def test_tables_follow_sigma_across_packets_and_windows(hip, oracle):
    p = synth.config1()
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    for sigma in (1.0, 1.0, 2.0, 2.0, 0.0, 1.0, 0.5, 0.5):      # repeats take the kept tables, changes rebuild them
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        c, g = fe.eval([0.1, -0.2, 0.05])
        c_ref, g_ref = ref.eval([0.1, -0.2, 0.05])
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, sigma
    w = synth.backend_window(20_000, 120, 90, 130.0, 130.0, 59.5, 44.5, 512, 256, 4, 7, 2, 0.2, seed=8)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    for sigma in (1.0, 1.0, 3.0, 1.0):
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, sigma, _lib.VARIANCE)
        ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, sigma, _lib.VARIANCE)
        ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, None)
        x = np.full(w.P, 0.003)
        c, g = be.eval(x)
        c_ref, g_ref = ref.eval(x)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, sigma
