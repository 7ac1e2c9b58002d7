"""-m gpu: CMX_OPT_SPLAT_MODE = 1 (LDS-privatised splat over events sorted by destination tile) vs the oracle.

The sort happens once per packet/window under the parameters of the first evaluation; later evaluations with other
parameters must stay exact (votes leaving a workgroup's LDS window take the global-atomic path) and a large drift
must trigger a re-binning."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    return synth.frontend_packet(60_011, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=12)


def _fe(hip, oracle, p, adjoint, measure=0, sigma=1.0):
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_splat_mode(1)
    if adjoint:
        fe.set_grad_mode(hip.GRAD_ADJOINT)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    return fe, ref


@pytest.mark.parametrize("adjoint", [True, False])
def test_frontend_lds_splat_parity_and_drift(hip, oracle, small, adjoint):
    fe, ref = _fe(hip, oracle, small, adjoint)
    # first evaluation bins under omega0; then walk away from it, including far outside the 16-px margin
    for om in ((0.3, -0.5, 0.2), (0.35, -0.55, 0.25), (0.6, -0.9, 0.4), (-4.0, 6.0, 3.0), (0, 0, 0)):
        c_ref, g_ref = ref.eval(om)
        assert rel_scalar(fe.eval(om, want_grad=False)[0], c_ref) < RTOL
        c, g = fe.eval(om)
        assert rel_scalar(c, c_ref) < RTOL
        assert rel_vec(g, g_ref) < RTOL
    st = fe.stats()
    assert st["rebins"] >= 2          # the jump to (-4, 6, 3) forced a re-sort
    assert st["fallback_frac"] < 0.15


def test_frontend_lds_iwe_image(hip, oracle, small):
    fe, ref = _fe(hip, oracle, small, True)
    om = (0.6, -0.9, 0.4)
    fe.eval(om, want_grad=False)
    assert rel_img(fe.computeImageOfWarpedEvents(om, blur=False), ref.iwe(om, blur=False)) < RTOL
    assert rel_img(fe.computeImageOfWarpedEvents(om, blur=True), ref.iwe(om, blur=True)) < RTOL


def test_frontend_lds_tiny_and_empty(hip, oracle, small):
    for n in (0, 1, 100, 257):
        p = synth.FrontendPacket(small.W, small.H, small.fx, small.fy, small.cx, small.cy, small.x[:n], small.y[:n],
                                 small.t_ns[:n], small.t_ref_ns, small.omega_true)
        fe, ref = _fe(hip, oracle, p, True)
        c_ref, g_ref = ref.eval((0.5, 0.5, 0.5))
        c, g = fe.eval((0.5, 0.5, 0.5))
        assert abs(c - c_ref) <= RTOL * max(abs(c_ref), 1e-12)
        assert np.abs(g - g_ref).max() <= RTOL * max(np.abs(g_ref).max(), 1e-12)


def test_frontend_config2_fast_path(hip, oracle):
    p = synth.config2()
    fe, ref = _fe(hip, oracle, p, True)
    for om in ((0.3, -0.5, 0.2), (0.5, -0.8, 0.35), (0.0, 0.0, 0.0)):
        c_ref, g_ref = ref.eval(om)
        c, g = fe.eval(om)
        assert rel_scalar(c, c_ref) < RTOL
        assert rel_vec(g, g_ref) < RTOL
        assert rel_scalar(fe.eval(om, want_grad=False)[0], c_ref) < RTOL


def _be(hip, oracle, w, IG=None, rate=1):
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_splat_mode(1)
    be.set_grad_mode(hip.GRAD_ADJOINT)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                  w.batch, rate, w.sigma, 0, IG)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, rate, w.sigma, 0)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    return be, ref


@pytest.mark.parametrize("order,K,nf,T", [(2, 5, 1, 0.2), (4, 10, 3, 0.35)])
def test_backend_lds_splat_parity_and_drift(hip, oracle, order, K, nf, T):
    w = synth.backend_window(60_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, order, K, nf, T, seed=15)
    be, ref = _be(hip, oracle, w)
    rng = np.random.default_rng(2)
    for d in (np.zeros(w.P), rng.normal(0, 0.005, w.P), rng.normal(0, 0.02, w.P), np.full(w.P, 0.5), np.full(w.P, 0.49)):
        c_ref, g_ref = ref.eval(d)
        assert rel_scalar(be.eval(d, want_grad=False)[0], c_ref) < RTOL
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL
        assert rel_vec(g, g_ref) < RTOL
        assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
        assert rel_img(be.get_plane(_lib.PLANE_IL_NEW), ref.IL_new) < RTOL
    assert be.stats()["rebins"] >= 2  # a 0.5 rad perturbation moves every vote out of its window: the next accumulate re-sorts


def test_backend_lds_sampling(hip, oracle):
    w = synth.backend_window(30_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 256, 128, 2, 5, 1, 0.2, seed=16)
    be, ref = _be(hip, oracle, w, rate=7)
    d = np.full(w.P, 0.003)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


def test_backend_config3_fast_path(hip, oracle):
    w = synth.config3()
    be, ref = _be(hip, oracle, w)
    d = np.zeros(w.P)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL
    assert rel_vec(g, g_ref) < RTOL
    # non-zero increments at full size: a point of a solve, then a large jump (re-sort of the tile order)
    for scale in (0.004, 0.05):
        d = np.random.default_rng(int(scale * 1000)).normal(0, scale, w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (scale, c, c_ref)
        assert rel_scalar(be.eval(d, False)[0], c_ref) < RTOL
