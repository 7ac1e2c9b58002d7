"""-m gpu: the packet / window pipeline entry points (cmx_frontend_prepare, cmx_backend_prepare) -- the tile sort, the streams and
the chunk table queued ahead of the first evaluation, at a HINT of the parameters.  Claims: results never depend on the hint
(only which votes find their LDS window does); a prepared packet's first evaluation does not sort again; a context can be
prepared for the next packet while another one is being solved, and the solve it then runs equals a fresh context's."""
import threading

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _fe(hip, p, det=False):
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    if det:
        fe.set_deterministic(True)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    return fe


@pytest.mark.parametrize("hint", [(0.0, 0.0, 0.0), (0.3, -0.5, 0.2), (9.0, -7.0, 5.0)])
def test_frontend_prepare_never_changes_results(hip, oracle, hint):
    p = synth.frontend_packet(120_000, 320, 240, 250.0, 250.0, 159.5, 119.5, seed=41)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    fe = _fe(hip, p)
    fe.prepare(hint)
    assert fe.stats()["rebins"] == 1
    om = np.array([0.3, -0.5, 0.2])
    c_ref, g_ref = ref.eval(om)
    c, g = fe.eval(om)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
    # a hint near the evaluation point: the first evaluation found the sort ready; a hint far away (an absurd 9 rad/s): votes land
    # beyond the reach of their tiles -- the two-launch evaluation (round 6: image pass inside the splat launch) notices, sorts
    # again at the evaluation point and repeats itself; results stay exact
    far = max(abs(h) for h in hint) > 5
    st = fe.stats()
    assert st["rebins"] == (2 if far else 1) and st["fused_redos"] == (1 if far else 0), st
    assert st["fallback_frac"] <= 0.03
    c2, g2 = fe.eval(om * 1.01)
    assert fe.stats()["rebins"] == (2 if far else 1)
    c2_ref, g2_ref = ref.eval(om * 1.01)
    assert rel_scalar(c2, c2_ref) < RTOL and rel_vec(g2, g2_ref) < RTOL


def test_prepare_next_packet_while_solving_this_one(hip):
    """Two contexts, two host threads: packet k is solved on A while packet k+1 is uploaded + prepared on B, then solved there.
    In CMX_OPT_DETERMINISTIC every solve equals, to the bit, the same packet solved alone on a fresh context."""
    packets = [synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=70 + k) for k in range(4)]
    alone = []
    for p in packets:
        fe = _fe(hip, p, det=True)
        alone.append(fe.setupProblemAndOptimize(np.zeros(3)))
        fe.close()
    ctxs = [hip.FrontendEvaluator(packets[0].W, packets[0].H, packets[0].lut) for _ in range(2)]
    for c in ctxs:
        c.set_fast_path()
        c.set_deterministic(True)

    def stage(c, p, hint):
        c.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        c.prepare(hint)
    stage(ctxs[0], packets[0], np.zeros(3))
    out = []
    for k, p in enumerate(packets):
        cur, nxt = ctxs[k % 2], ctxs[(k + 1) % 2]
        th = None
        if k + 1 < len(packets):
            th = threading.Thread(target=stage, args=(nxt, packets[k + 1], out[-1][0] if out else np.zeros(3)))
            th.start()
        out.append(cur.setupProblemAndOptimize(np.zeros(3)))
        if th:
            th.join()
    for (x, rep), (x0, rep0) in zip(out, alone):
        assert np.array_equal(x, x0), (x, x0)
        assert rep["final_cost"] == rep0["final_cost"] and rep["n_f"] == rep0["n_f"] and rep["n_df"] == rep0["n_df"]
    for c in ctxs:
        c.close()


def test_backend_prepare_never_changes_results(hip, oracle):
    w = synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 3, 0.25, seed=19)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = np.random.default_rng(3).normal(0, 0.004, w.P)
    c_ref, g_ref = ref.eval(d)
    for hint in (None, d, np.full(w.P, 0.4)):
        be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        be.set_fast_path()
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, w.sigma, _lib.VARIANCE)
        be.prepare(hint)
        assert be.stats()["rebins"] == 1
        c, g = be.eval(d)
        assert be.stats()["rebins"] == 1   # the first evaluation did not sort again
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, hint
        assert rel_scalar(be.eval(d, False)[0], c_ref) < RTOL
        be.close()


def test_tiny_image_sigma_change_keeps_the_operator_tables_consistent(hip, oracle):
    """an image with one side <= 4r builds the banded G^T G table of ONE axis only; the composite image pass must
    then not pair it with the other axis' table of an earlier, smaller radius."""
    p = synth.frontend_packet(3_000, 18, 64, 20.0, 20.0, 8.5, 31.5, seed=5)
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    om = np.array([0.2, -0.3, 0.1])
    for sigma in (0.5, 1.0, 1.25, 0.5, 1.25):   # radii 2, 4, 5: at r = 5 the 18-pixel axis has no table (18 <= 4*5)
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, oracle.VARIANCE)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        c_ref, g_ref = ref.eval(om)
        c, g = fe.eval(om)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, sigma


def test_stats_take_their_length(hip):
    import ctypes as C
    assert _lib.lib().cmx_abi_version() == 6
    p = synth.frontend_packet(1_000, 64, 48, 60.0, 60.0, 31.5, 23.5, seed=1)
    fe = _fe(hip, p)
    fe.eval(np.zeros(3))
    buf = (C.c_double * 8)(*([-7.0] * 8))
    assert _lib.lib().cmx_get_stats(fe._ctx, buf, 4) == 0
    assert buf[0] == 1.0 and buf[3] == 1000.0 and all(buf[k] == -7.0 for k in range(4, 8))   # nothing past the caller's length
