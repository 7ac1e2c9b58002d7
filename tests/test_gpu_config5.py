"""-m gpu: BASELINE config 5 -- 1 Mpix sensor (1280x720), equirectangular 4096x2048 global map, the later-window
shape of the launch defaults (linear spline, K = 5, no fixed control pose => P = 15; SURVEY.md section 8 B-note) and a
non-zero global map, so that alpha != 0 and I = IL + alpha*IGp are exercised at the full map size.

  * parity vs the CPU oracle on a 150k-event shard (the oracle needs seconds for the sixteen 8.4-Mpix planes);
  * at the per-GPU size of the 8-GPU configuration (2.5M events) through size-independent properties: two shards
    accumulate to the whole, the fast path equals the reference-shaped path, central differences match the gradient."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _set(be, w, IG, x=None, y=None, t=None):
    x, y, t = (w.x, w.y, w.t_ns) if x is None else (x, y, t)
    be.set_window(x, y, t, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE, IG)


@pytest.fixture(scope="module")
def small():
    return synth.config5(N=150_000)


@pytest.fixture(scope="module")
def prior_map(hip, small):
    """A previous window's contribution to the global map: IL_old of a neighbouring window, scaled."""
    prev = synth.config5(N=120_000, seed=synth.SEED0 + 55)
    be = hip.reference_shaped.BackendEvaluator(prev.W, prev.H, prev.lut, prev.Wp, prev.Hp)
    _set(be, prev, None)
    be.eval(np.zeros(prev.P), False)
    return np.ascontiguousarray(be.get_plane(_lib.PLANE_IL_OLD) * 2.5)


def test_config5_parity_with_oracle(hip, oracle, small, prior_map):
    w = small
    assert (w.W, w.H, w.Wp, w.Hp, w.order, w.K, w.num_fixed, w.P) == (1280, 720, 4096, 2048, 2, 5, 0, 15)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, prior_map)
    d = np.random.default_rng(8).normal(0, 0.004, w.P)
    c_ref, g_ref = ref.eval(d)
    assert ref.alpha > 0
    for fast in (False, True):
        be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        if fast:
            be.set_fast_path()
        _set(be, w, prior_map)
        c, g = be.eval(d)
        assert rel_scalar(be.alpha, ref.alpha) < RTOL
        assert rel_scalar(c, c_ref) < RTOL, (fast, c, c_ref)
        assert rel_vec(g, g_ref) < RTOL, (fast, g, g_ref)
        assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
        c0, _ = be.eval(d, False)
        assert rel_scalar(c0, c_ref) < RTOL


def test_config5_per_gpu_size_properties(hip, prior_map):
    w = synth.config5(N=2_500_000, seed=synth.SEED0 + 56)
    fast = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    fast.set_fast_path()
    fast.set_option(_lib.OPT_REUSE_IMAGE, 0)
    _set(fast, w, prior_map)
    d = np.random.default_rng(9).normal(0, 0.003, w.P)
    c, g = fast.eval(d)
    alpha = fast.alpha
    assert alpha > 0 and np.isfinite(c) and np.all(np.isfinite(g))
    # (1) the fast path equals the reference-shaped path (derivative planes, one global atomic per vote)
    slow = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    _set(slow, w, prior_map)
    c2, g2 = slow.eval(d)
    assert rel_scalar(slow.alpha, alpha) < RTOL and rel_scalar(c2, c) < RTOL and rel_vec(g2, g) < RTOL
    # (2) two shards of whole batches accumulate to the whole window (what the all-reduce across GPUs relies on)
    il_old, il_new = fast.get_plane(_lib.PLANE_IL_OLD), fast.get_plane(_lib.PLANE_IL_NEW)
    cut = (len(w.x) // 2) // w.batch * w.batch
    parts = []
    for sl in (slice(0, cut), slice(cut, None)):
        h = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        h.set_fast_path()
        _set(h, w, None, w.x[sl], w.y[sl], w.t_ns[sl])
        h.accumulate(d, False)
        parts.append((h.get_plane(_lib.PLANE_IL_OLD), h.get_plane(_lib.PLANE_IL_NEW)))
    assert rel_img(parts[0][0] + parts[1][0], il_old) < RTOL
    assert rel_img(parts[0][1] + parts[1][1], il_new) < RTOL
    # (3) the analytic gradient is the derivative of the cost (alpha frozen after the first evaluation): directional
    #     central difference along the gradient, fp32-accumulated cost => step large enough to beat its noise
    u = g / np.linalg.norm(g)
    h = 2e-3
    cp, _ = fast.eval(d + h * u, False)
    cm, _ = fast.eval(d - h * u, False)
    fd = (cp - cm) / (2 * h)
    assert abs(fd - float(g @ u)) < 0.02 * abs(float(g @ u)), (fd, float(g @ u))
