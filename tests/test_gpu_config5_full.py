"""-m gpu: BASELINE config 5 AT ITS OWN SIZE on one GPU -- 1280x720 sensor, 20M events (8 time slabs of 2.5M, exactly what
the ranks of the 8-GPU run generate and own), equirectangular 4096x2048 global map, linear spline K = 5, no fixed control
pose (P = 15), a non-zero global map so that alpha != 0 (event_pano_warper.cpp:201-213; launch/ecrot_handheld.launch:34).
Mirror of tests/test_gpu_baseline_configs.py::test_config4_whole_window_on_one_gpu:

  * the production path on all 20M events against the CPU oracle on all 20M events (contrast, gradient, alpha, both
    vote planes), at a non-zero increment;
  * the production path against the reference-shaped GPU path (derivative planes, one global atomic per vote);
  * the eight dist.batch_range shards accumulating to the whole window (what the all-reduce relies on);
  * one shard on its own against the oracle."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, dist, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu
WORLD, PER_GPU = 8, 2_500_000


def _set(be, w, IG, x=None, y=None, t=None):
    x, y, t = (w.x, w.y, w.t_ns) if x is None else (x, y, t)
    be.set_window(x, y, t, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE, IG)


@pytest.fixture(scope="module")
def c5():
    slabs = [synth.config5_slab(r, WORLD, PER_GPU) for r in range(WORLD)]
    return slabs, synth.concat_slabs(slabs)


@pytest.fixture(scope="module")
def prior_map(hip):
    """A previous window's contribution to the global map: IL_old of a neighbouring window, scaled."""
    prev = synth.config5(N=400_000, seed=synth.SEED0 + 55)
    be = hip.BackendEvaluator(prev.W, prev.H, prev.lut, prev.Wp, prev.Hp)
    be.set_fast_path()
    _set(be, prev, None)
    be.eval(np.zeros(prev.P), False)
    IG = np.ascontiguousarray(be.get_plane(_lib.PLANE_IL_OLD) * 6.0)
    be.close()
    return IG


def test_config5_whole_window_on_one_gpu(hip, oracle, c5, prior_map):
    slabs, w = c5
    assert len(w.x) == 20_000_000
    assert (w.W, w.H, w.Wp, w.Hp, w.order, w.K, w.num_fixed, w.P) == (1280, 720, 4096, 2048, 2, 5, 0, 15)
    d = np.random.default_rng(51).normal(0, 0.003, w.P)
    fast = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    fast.set_fast_path()
    fast.set_option(_lib.OPT_REUSE_IMAGE, 0)
    _set(fast, w, prior_map)
    c, g = fast.eval(d)
    alpha = fast.alpha
    il_old, il_new = fast.get_plane(_lib.PLANE_IL_OLD), fast.get_plane(_lib.PLANE_IL_NEW)
    # (1) the CPU oracle on all 20M events (sixteen 8.4-Mpix planes)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, prior_map)
    c_ref, g_ref = ref.eval(d)
    assert ref.alpha > 0 and rel_scalar(alpha, ref.alpha) < RTOL, (alpha, ref.alpha)
    assert rel_scalar(c, c_ref) < RTOL, (c, c_ref)
    assert rel_vec(g, g_ref) < RTOL, (g, g_ref)
    assert rel_img(il_old, ref.IL_old) < RTOL and rel_img(il_new, ref.IL_new) < RTOL
    assert rel_scalar(fast.eval(d, False)[0], c_ref) < RTOL
    # a second point of the same window (alpha stays frozen, event_pano_warper.cpp:201)
    d2 = np.random.default_rng(52).normal(0, 0.01, w.P)
    c2_ref, g2_ref = ref.eval(d2)
    c2, g2 = fast.eval(d2)
    assert rel_scalar(fast.alpha, ref.alpha) < RTOL
    assert rel_scalar(c2, c2_ref) < RTOL and rel_vec(g2, g2_ref) < RTOL, (c2, c2_ref, g2, g2_ref)
    # (2) the reference-shaped GPU path (2 + 15 planes of 32 MB, one global atomic per vote)
    slow = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    _set(slow, w, prior_map)
    cs, gs = slow.eval(d)
    assert rel_scalar(slow.alpha, ref.alpha) < RTOL
    slow.close()
    assert rel_scalar(cs, c_ref) < RTOL and rel_vec(gs, g_ref) < RTOL
    # (3) the eight batch-range shards of the 8-GPU run accumulate to the whole window
    acc_old, acc_new = np.zeros_like(il_old, dtype=np.float64), np.zeros_like(il_new, dtype=np.float64)
    h = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    h.set_fast_path()
    for r in range(WORLD):
        beg, end = dist.batch_range(len(w.x), w.batch, r, WORLD)
        assert (beg, end) == (r * PER_GPU, (r + 1) * PER_GPU)   # a rank's shard is exactly its time slab
        assert np.array_equal(w.x[beg:end], slabs[r].x) and np.array_equal(w.t_ns[beg:end], slabs[r].t_ns)
        _set(h, w, None, w.x[beg:end], w.y[beg:end], w.t_ns[beg:end])
        h.accumulate(d, False)
        acc_old += h.get_plane(_lib.PLANE_IL_OLD)
        acc_new += h.get_plane(_lib.PLANE_IL_NEW)
    assert rel_img(acc_old, il_old) < RTOL and rel_img(acc_new, il_new) < RTOL


@pytest.mark.parametrize("rank", [0, 5])
def test_config5_one_rank_shard_vs_oracle(hip, oracle, c5, prior_map, rank):
    slabs, w = c5
    s = slabs[rank]
    d = np.random.default_rng(60 + rank).normal(0, 0.004, w.P)
    be = hip.BackendEvaluator(s.W, s.H, s.lut, s.Wp, s.Hp)
    be.set_fast_path()
    _set(be, s, prior_map)
    ref = oracle.Backend(s.W, s.H, s.lut, s.Wp, s.Hp, s.order, s.batch, s.sample_rate, s.sigma, oracle.VARIANCE)
    ref.set_window(s.x, s.y, s.t_ns, s.knots_init, s.start_ns, s.dt_ns, s.num_fixed, s.t_next_win_beg_ns, prior_map)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(be.alpha, ref.alpha) < RTOL
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IL_NEW), ref.IL_new) < RTOL
