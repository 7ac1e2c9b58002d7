"""-m gpu: the back end's image passes skip panorama tiles with nothing within the filter's reach and clear only the
dirty tiles of the ping-pong partner (tile-occupancy flags).  Everything observable must stay
what the oracle computes over the full image: sequences of evaluations whose votes move between tiles, global-map
content away from the events, blur radii beyond one tile row, switches between the fast and the reference-shaped path."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def w():
    return synth.backend_window(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 1024, 512, 2, 5, 1, 0.2, seed=71)


def _ig(w, kind):
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    yy, xx = np.mgrid[0:w.Hp, 0:w.Wp]
    if kind in ("far", "both"):      # a blob in a corner the camera never looks at, touching the image border
        IG += (3.0 * np.exp(-((xx - 20) ** 2 + (yy - 500) ** 2) / 200.0)).astype(np.float32) * ((xx < 60) & (yy > 440))
    if kind in ("near", "both"):     # and one under the events (panorama centre)
        IG += (2.0 * np.exp(-((xx - 520) ** 2 + (yy - 250) ** 2) / 900.0)).astype(np.float32) * (np.hypot(xx - 520, yy - 250) < 90)
    return IG


def _pair(hip, oracle, w, IG, sigma=1.0):
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, sigma, _lib.VARIANCE, IG)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    return be, ref


@pytest.mark.parametrize("kind", ["none", "far", "both"])
def test_moving_votes_and_map_content(hip, oracle, w, kind):
    IG = None if kind == "none" else _ig(w, kind)
    be, ref = _pair(hip, oracle, w, IG)
    rng = np.random.default_rng(2)
    big = np.tile([0.0, 0.25, 0.0], w.P // 3)           # a 14-degree yaw: every vote lands ~40 px away, in other tiles
    seq = [np.zeros(w.P), big, rng.normal(0, 0.01, w.P), -big, np.zeros(w.P)]
    for i, d in enumerate(seq):
        want_grad = i % 2 == 0
        c_ref, g_ref = ref.eval(d, want_grad)
        c, g = be.eval(d, want_grad)
        assert rel_scalar(c, c_ref) < RTOL, (kind, i, c, c_ref)
        if want_grad:
            assert rel_vec(g, g_ref) < RTOL, (kind, i)
        # the accumulation planes hold exactly this evaluation's votes: nothing survived the selective clearing
        assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
        assert rel_img(be.get_plane(_lib.PLANE_IL_NEW), ref.IL_new) < RTOL
    if IG is not None:
        assert rel_scalar(be.alpha, ref.alpha) < RTOL and be.alpha > 0
    # df right after f at the same point (image reuse) on flagged planes
    d = rng.normal(0, 0.02, w.P)
    c0, _ = be.eval(d, False)
    c1, g1 = be.eval(d, True)
    c_ref, g_ref = ref.eval(d, True)
    assert rel_scalar(c0, c_ref) < RTOL and rel_scalar(c1, c_ref) < RTOL and rel_vec(g1, g_ref) < RTOL


@pytest.mark.parametrize("sigma", [2.0, 3.0])
def test_blur_reach_beyond_one_tile_row(hip, oracle, w, sigma):
    """radius 8 / 12: the adjoint pass needs the image within 2r = 16 / 24 rows, i.e. two tile rows away."""
    be, ref = _pair(hip, oracle, w, _ig(w, "both"), sigma=sigma)
    for d in (np.zeros(w.P), np.random.default_rng(4).normal(0, 0.01, w.P)):
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
        assert rel_scalar(be.eval(d, False)[0], c_ref) < RTOL


def test_switching_between_fast_and_reference_shaped_paths(hip, oracle, w):
    be, ref = _pair(hip, oracle, w, _ig(w, "near"))
    rng = np.random.default_rng(6)
    for step in range(6):
        d = rng.normal(0, 0.02, w.P)
        if step % 3 == 1:      # derivative planes + global atomics: writes the planes without occupancy flags
            be.set_grad_mode(_lib.GRAD_PLANES)
            be.set_splat_mode(0)
        else:
            be.set_fast_path()
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, step
        assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
    # the full-image outputs (display path) are not subject to tile skipping
    iwe = be.computeImageOfWarpedEvents(d)
    iwe_ref = ref.iwe(d)
    assert rel_img(iwe, iwe_ref) < RTOL


def test_empty_window_and_empty_map(hip, oracle, w):
    """No accepted vote at all (every event warped out of the panorama's accepted band is impossible on a sphere, so
    use zero events): contrast 0, gradient 0, no NaN from skipped tiles."""
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    e = np.zeros(0, np.uint16)
    be.set_window(e, e, np.zeros(0, np.int64), w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed,
                  w.t_next_win_beg_ns)
    c, g = be.eval(np.zeros(w.P))
    assert c == 0.0 and np.all(g == 0.0)


def test_panorama_beyond_the_counting_sort(hip, oracle):
    """8192 x 4096: 2 x 32768 destination tiles, more keys than the counting sort's LDS histogram holds, so the tile
    sort takes the radix-sort route; same answers as the oracle (cost) and as the splat that needs no sort (gradient)."""
    w = synth.backend_window(20_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 8192, 4096, 2, 5, 1, 0.2, seed=72)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE, None)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, None)
    d = 0.01 * np.cos(np.arange(w.P))
    c_ref, _ = ref.eval(d, want_grad=False)
    c, g = be.eval(d)
    assert be.stats()["rebins"] == 1
    assert rel_scalar(c, c_ref) < RTOL
    be.set_splat_mode(0)  # one global atomic per vote: no tile sort at all
    c0, g0 = be.eval(d)
    assert rel_scalar(c0, c_ref) < RTOL and rel_vec(g, g0) < RTOL
