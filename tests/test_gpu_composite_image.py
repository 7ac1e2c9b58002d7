"""-m gpu: CMX_OPT_COMPOSITE_IMAGE -- the adjoint image pass with G^T G applied as one banded operator per axis (three
phases per tile instead of five; blur radius 4 = the reference's blur_sigma 1).  Against the CPU oracle and against the
four-pass form, on images whose border tiles dominate / that are smaller than a tile / not a multiple of the tile, for
three sigmas of radius 4, both contrast measures; back end with a prior map (alpha != 0) on a panorama walked through
the tile list."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _fe(hip, p, sigma, measure, composite):
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_option(_lib.OPT_COMPOSITE_IMAGE, 1 if composite else 0)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
    return fe


@pytest.mark.parametrize("sigma", [0.95, 1.0, 1.05])       # all radius 4, different taps
@pytest.mark.parametrize("W,H", [(240, 180), (70, 50), (333, 97), (40, 18), (20, 16), (640, 480)])
def test_frontend_composite_image_pass_matches_oracle_and_four_pass_form(hip, oracle, W, H, sigma):
    f = 0.9 * max(W, H)
    p = synth.frontend_packet(30_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, seed=71)
    for measure in (_lib.VARIANCE, _lib.MEAN_SQUARE):
        a, b = _fe(hip, p, sigma, measure, True), _fe(hip, p, sigma, measure, False)
        ref = oracle.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        for om in ([0.3, -0.5, 0.2], p.omega_true, [-2.0, 1.5, 3.0], [0.0, 0.0, 0.0]):
            c_ref, g_ref = ref.eval(om)
            ca, ga = a.eval(om)
            cb, gb = b.eval(om)
            assert rel_scalar(ca, c_ref) < RTOL and rel_vec(ga, g_ref) < RTOL, (W, H, sigma, measure, om, ca, c_ref, ga, g_ref)
            assert rel_scalar(ca, cb) < 1e-6 and rel_vec(ga, gb) < 2e-6, (W, H, sigma, measure, om, ga, gb)
            # df after f at the same point: Jt left by the speculative pass of the cost-only evaluation
            a.set_option(_lib.OPT_REUSE_IMAGE, 1)
            cf, _ = a.eval(np.asarray(om) * 0.9, False)
            _, gf = a.eval(np.asarray(om) * 0.9, True)
            c9, g9 = ref.eval(np.asarray(om) * 0.9)
            assert rel_scalar(cf, c9) < RTOL and rel_vec(gf, g9) < RTOL


def test_composite_tables_follow_sigma(hip, oracle):
    """set_packet with another sigma rebuilds (radius <= 6) or drops (radius > 6) the operator tables."""
    p = synth.config1()
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    for sigma in (1.0, 2.0, 1.0, 0.5, 1.05):
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        c, g = fe.eval([0.1, -0.2, 0.05])
        c_ref, g_ref = ref.eval([0.1, -0.2, 0.05])
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (sigma, g, g_ref)


@pytest.mark.parametrize("Wp,Hp", [(512, 256), (1000, 300), (4096, 2048)])   # one workgroup per tile / tile-list walk
def test_backend_composite_image_pass(hip, oracle, Wp, Hp):
    W, H = 120, 90
    f = 1.1 * W
    w = synth.backend_window(25_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, Wp, Hp, 4, 7, 2, 0.2, seed=333)
    rng = np.random.default_rng(5)
    IG = np.zeros((Hp, Wp), np.float32)
    IG[Hp // 2 - 20:Hp // 2 + 20, Wp // 2 - 40:Wp // 2 + 40] = rng.uniform(0.5, 3.0, (40, 80)).astype(np.float32)
    evs = []
    for composite in (1, 0):
        be = hip.BackendEvaluator(W, H, w.lut, Wp, Hp)
        be.set_fast_path()
        be.set_option(_lib.OPT_COMPOSITE_IMAGE, composite)
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, 1.0, _lib.VARIANCE, IG)
        evs.append(be)
    ref = oracle.Backend(W, H, w.lut, Wp, Hp, w.order, w.batch, w.sample_rate, 1.0, _lib.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    for x in (np.zeros(w.P), rng.normal(0, 0.01, w.P), rng.normal(0, 0.05, w.P)):
        c_ref, g_ref = ref.eval(x)
        ca, ga = evs[0].eval(x)
        cb, gb = evs[1].eval(x)
        assert rel_scalar(ca, c_ref) < RTOL and rel_vec(ga, g_ref) < RTOL, (Wp, Hp, ca, c_ref)
        assert rel_scalar(ca, cb) < 1e-6 and rel_vec(ga, gb) < 5e-6
    assert evs[0].alpha != 0.0


@pytest.mark.parametrize("sigma", [0.5, 1.7, 2.0, 3.0])       # radius 2, 7, 8, 12: the run-time-radius form of the composite pass
@pytest.mark.parametrize("Wp,Hp", [(600, 300), (4096, 2048)])  # one workgroup per tile / tile-list walk
def test_backend_composite_image_pass_other_radii(hip, oracle, Wp, Hp, sigma):
    W, H = 120, 90
    f = 1.1 * W
    w = synth.backend_window(25_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, Wp, Hp, 4, 7, 2, 0.2, seed=334)
    rng = np.random.default_rng(6)
    IG = np.zeros((Hp, Wp), np.float32)
    IG[Hp // 2 - 20:Hp // 2 + 20, Wp // 2 - 40:Wp // 2 + 40] = rng.uniform(0.5, 3.0, (40, 80)).astype(np.float32)
    evs = []
    for composite in (1, 0):
        be = hip.BackendEvaluator(W, H, w.lut, Wp, Hp)
        be.set_fast_path()
        be.set_option(_lib.OPT_COMPOSITE_IMAGE, composite)
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, sigma, _lib.VARIANCE, IG)
        evs.append(be)
    ref = oracle.Backend(W, H, w.lut, Wp, Hp, w.order, w.batch, w.sample_rate, sigma, _lib.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    for x in (np.zeros(w.P), rng.normal(0, 0.02, w.P)):
        c_ref, g_ref = ref.eval(x)
        ca, ga = evs[0].eval(x)
        cb, gb = evs[1].eval(x)
        assert rel_scalar(ca, c_ref) < RTOL and rel_vec(ga, g_ref) < RTOL, (Wp, Hp, sigma)
        assert rel_scalar(ca, cb) < 1e-6 and rel_vec(ga, gb) < RTOL


@pytest.mark.parametrize("W,H,sigma", [(60, 50, 3.0), (49, 53, 3.0), (48, 60, 3.0), (100, 35, 2.0), (33, 33, 2.0)])
def test_frontend_composite_small_images_large_radius(hip, oracle, W, H, sigma):
    """image barely larger (or not larger) than the operator's 4r+1 band: tables built iff both sides exceed 4r"""
    f = 0.9 * max(W, H)
    p = synth.frontend_packet(8_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, seed=72)
    fe = _fe(hip, p, sigma, _lib.VARIANCE, True)
    ref = oracle.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, _lib.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for om in ([0.3, -0.5, 0.2], p.omega_true):
        c_ref, g_ref = ref.eval(om)
        c, g = fe.eval(om)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (W, H, sigma, g, g_ref)
