"""-m gpu: back-end (panoramic BA) HIP path vs the CPU oracle, through the C ABI.

Mirrors global_contrast_fdf's data flow (global_optim_contrast_gsl_analytical.cpp:17-68): incremental rotation
vectors -> CopyAndIncrementalUpdate -> EventWarper::computeImageOfWarpedEvents -> computeContrast."""
import numpy as np
import pytest

from cmax_slam_amd import synth
from cmax_slam_amd import _lib
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _pair(hip, oracle, w, measure=0, sigma=None, batch=None, rate=None, IG=None, knots=None):
    sigma = w.sigma if sigma is None else sigma
    batch = w.batch if batch is None else batch
    rate = w.sample_rate if rate is None else rate
    knots = w.knots_init if knots is None else knots
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_window(w.x, w.y, w.t_ns, w.order, knots, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                  batch, rate, sigma, measure, IG)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, batch, rate, sigma, measure)
    ref.set_window(w.x, w.y, w.t_ns, knots, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    return be, ref


def _window(order, K, nf, T, N=40_003, seed=5, Wp=512, Hp=256):
    return synth.backend_window(N, 240, 180, 200.0, 200.0, 119.5, 89.5, Wp, Hp, order, K, nf, T, seed=seed)


@pytest.fixture(scope="module")
def cubic():
    return _window(4, 10, 3, 0.35)


@pytest.fixture(scope="module")
def linear():
    return _window(2, 5, 1, 0.2)


@pytest.mark.parametrize("which", ["linear", "cubic"])
def test_planes_and_derivative_planes(hip, oracle, which, linear, cubic):
    w = linear if which == "linear" else cubic
    be, ref = _pair(hip, oracle, w)
    rng = np.random.default_rng(3)
    drot = rng.normal(0, 0.01, w.P)
    iwe, planes = be.computeImageOfWarpedEvents(drot, want_deriv=True)
    iwe_ref, planes_ref = ref.iwe(drot, planes=True)
    assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IL_NEW), ref.IL_new) < RTOL
    assert ref.IL_old.sum() > 0 and ref.IL_new.sum() > 0  # both halves of the window are populated
    assert rel_img(iwe, iwe_ref) < RTOL
    assert planes.shape == planes_ref.shape
    scale = np.abs(planes_ref).max()
    for j in range(w.P):
        assert np.abs(planes[j].astype(np.float64) - planes_ref[j]).max() < RTOL * scale


@pytest.mark.parametrize("which", ["linear", "cubic"])
@pytest.mark.parametrize("measure", [0, 1])
def test_contrast_fdf(hip, oracle, which, measure, linear, cubic):
    w = linear if which == "linear" else cubic
    be, ref = _pair(hip, oracle, w, measure=measure)
    rng = np.random.default_rng(4)
    for drot in (np.zeros(w.P), rng.normal(0, 0.01, w.P)):
        c_ref, g_ref = ref.eval(drot)
        f, df = be.contrast_fdf(drot)
        assert rel_scalar(-f, c_ref) < RTOL
        assert rel_vec(-df, g_ref) < RTOL
        assert rel_scalar(-be.contrast_f(drot), c_ref) < RTOL


def test_global_map_and_alpha(hip, oracle, linear):
    """Non-zero IG: alpha is computed by the first evaluation of the window and frozen
    (event_pano_warper.cpp:201-213, updateAlpha :134-165)."""
    w = linear
    # a previous window's map: IL_old of the true trajectory, from the oracle
    be0, ref0 = _pair(hip, oracle, w, knots=w.knots_true)
    ref0.iwe(np.zeros(w.P))
    IG = ref0.IL_old.copy() * 1.7
    be, ref = _pair(hip, oracle, w, IG=IG)
    rng = np.random.default_rng(8)
    d0, d1 = np.zeros(w.P), rng.normal(0, 0.01, w.P)
    c_ref0, g_ref0 = ref.eval(d0)
    c0, g0 = be.eval(d0)
    assert ref.alpha > 0
    assert rel_scalar(be.alpha, ref.alpha) < RTOL
    assert rel_scalar(c0, c_ref0) < RTOL and rel_vec(g0, g_ref0) < RTOL
    a_first = be.alpha
    c_ref1, g_ref1 = ref.eval(d1)  # alpha stays what the first evaluation made it
    c1, g1 = be.eval(d1)
    assert be.alpha == a_first
    assert rel_scalar(c1, c_ref1) < RTOL and rel_vec(g1, g_ref1) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IWE), ref.iwe(d1)) < RTOL


@pytest.mark.parametrize("n,batch,rate", [(2, 100, 1), (101, 100, 1), (201, 100, 1), (300, 100, 1), (1000, 64, 3),
                                          (40_003, 100, 7), (5001, 1000, 1000)])
def test_batching_quirks_and_sampling(hip, oracle, linear, n, batch, rate):
    """for (beg; beg < end-1; beg += B): a trailing batch holding exactly one event is skipped (:188-196);
    the sampling stride restarts at every batch start (:262)."""
    w = linear
    w2 = synth.BackendWindow(w.W, w.H, w.fx, w.fy, w.cx, w.cy, w.Wp, w.Hp, w.order, w.x[:n], w.y[:n], w.t_ns[:n],
                             w.knots_true, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    be, ref = _pair(hip, oracle, w2, batch=batch, rate=rate)
    d = np.full(w.P, 0.003)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert abs(c - c_ref) <= RTOL * max(abs(c_ref), 1e-12)
    assert np.abs(g - g_ref).max() <= RTOL * max(np.abs(g_ref).max(), 1e-12)
    assert rel_img(be.get_plane(_lib.PLANE_IL_OLD) + be.get_plane(_lib.PLANE_IL_NEW), ref.IL_old + ref.IL_new) < RTOL \
        or ref.IL_old.sum() + ref.IL_new.sum() == 0


def test_spline_support_is_checked(hip, cubic):
    w = cubic
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    with pytest.raises(hip.CmaxHipError) as e:  # too few knots for the window: Basalt would assert (so3_spline.h:227-230)
        be.set_window(w.x, w.y, w.t_ns, 4, w.knots_init[:6], w.start_ns, w.dt_ns, 3, w.t_next_win_beg_ns)
    assert e.value.status == 4
    with pytest.raises(hip.CmaxHipError) as e:  # events before the spline start (:221)
        be.set_window(w.x, w.y, w.t_ns, 4, w.knots_init, w.start_ns + 10_000_000, w.dt_ns, 3, w.t_next_win_beg_ns)
    assert e.value.status == 4
    with pytest.raises(hip.CmaxHipError) as e:
        be.set_window(w.x, w.y, w.t_ns, 3, w.knots_init, w.start_ns, w.dt_ns, 3, w.t_next_win_beg_ns)
    assert e.value.status == 1


def test_zero_fixed_later_window_shape(hip, oracle):
    """Later windows of the launch defaults: linear, K=5, 0 fixed => P=15 (pose_graph_optimizer.cpp:283-288)."""
    w = _window(2, 5, 0, 0.2, seed=9)
    be, ref = _pair(hip, oracle, w)
    d = np.random.default_rng(1).normal(0, 0.01, w.P)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert w.P == 15
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL


def test_config3_full_size(hip, oracle):
    """BASELINE config 3: cubic, K=10 (P=21), 5M events, 1024x1024 panorama vs the oracle."""
    w = synth.config3()
    be, ref = _pair(hip, oracle, w)
    d = np.zeros(w.P)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL
    assert rel_vec(g, g_ref) < RTOL
    assert rel_scalar(be.eval(d, want_grad=False)[0], c_ref) < RTOL
    # mass conservation at full size
    tot = be.get_plane(_lib.PLANE_IL_OLD).sum(dtype=np.float64) + be.get_plane(_lib.PLANE_IL_NEW).sum(dtype=np.float64)
    tot_ref = ref.IL_old.sum(dtype=np.float64) + ref.IL_new.sum(dtype=np.float64)
    assert abs(tot - tot_ref) < 1e-5 * len(w.x)
    # ... and at non-zero increments: a point of a solve (small) and a large one on every free knot
    for scale in (0.004, 0.05):
        d = np.random.default_rng(int(scale * 1000)).normal(0, scale, w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (scale, c, c_ref)
        assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL


@pytest.mark.parametrize("order,K,nf,T", [(2, 16, 1, 0.75), (4, 16, 3, 0.65), (2, 17, 1, 0.8), (4, 24, 3, 1.05)])
def test_both_pose_table_forms(hip, oracle, order, K, nf, T):
    """K <= 16: the pose-table kernel takes log / J^-1 / R(knot^-1) of every knot pair from the host (once per
    evaluation); more knots: the self-contained form.  Same values either way (so3_spline.h:218-274), checked through
    the planes, the contrast and the gradient in both gradient modes, with large increments on every knot."""
    w = _window(order, K, nf, T, N=30_000, seed=13)
    be, ref = _pair(hip, oracle, w)
    d = np.random.default_rng(K).normal(0, 0.05, w.P)
    c_ref, g_ref = ref.eval(d)
    for path in ("fast", "reference"):
        be.set_fast_path() if path == "fast" else be.set_reference_path()
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (path, K)
        assert rel_scalar(be.eval(d, want_grad=False)[0], c_ref) < RTOL
    np.testing.assert_allclose(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old, atol=2e-4)
