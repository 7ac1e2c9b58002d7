"""-m gpu: device-resident global-map upkeep (EventWarper::updateIG / setUpdateTimesIG, event_pano_warper.cpp:81-126)
against the oracle, over a two-window sequence: solve window 1, fold IL_old into IG where the visit count allows,
mark the visited area, then evaluate window 2 against the RESIDENT map (CMX_KEEP_MAP) -- alpha and contrast must match
an oracle that carried the same map on the host."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def test_two_windows_with_resident_map(hip, oracle):
    w1 = synth.backend_window(40_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 1, 0.2, seed=21)
    w2 = synth.backend_window(40_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 0, 0.2, seed=21, noise=0.2)
    be = hip.BackendEvaluator(w1.W, w1.H, w1.lut, w1.Wp, w1.Hp)
    be.set_fast_path()
    ref = oracle.Backend(w1.W, w1.H, w1.lut, w1.Wp, w1.Hp, 2)

    # ---- window 1 (empty map)
    be.set_window(w1.x, w1.y, w1.t_ns, 2, w1.knots_init, w1.start_ns, w1.dt_ns, w1.num_fixed, w1.t_next_win_beg_ns)
    ref.set_window(w1.x, w1.y, w1.t_ns, w1.knots_init, w1.start_ns, w1.dt_ns, w1.num_fixed, w1.t_next_win_beg_ns)
    d = np.full(w1.P, 0.002)
    c, _ = be.eval(d, False)          # "the last evaluation performed" defines IL_old (pose_graph_optimizer.cpp:303)
    c_ref, _ = ref.eval(d, False)
    assert rel_scalar(c, c_ref) < RTOL

    # visit bookkeeping first for two poses, then a second mark so some counts reach 2
    qs = [w1.knots_true[0], w1.knots_true[2], w1.knots_true[2]]
    for q in qs:
        be.setUpdateTimesIG(q, 3)
        ref.mark_visited(q, 3)
    be.updateIG(1)                    # max_update_times = 1: pixels visited twice stop accumulating
    ref.update_ig(1)
    IG, visits = be.getIG(with_visits=True)
    np.testing.assert_array_equal(visits, ref.update_times)
    assert visits.max() == 3 and (visits == 0).any()
    assert rel_img(IG, ref.IG) < RTOL
    assert 0 < IG.sum() < ref.IL_old.sum()   # some pixels were frozen by the visit count

    # ---- window 2 against the resident map
    be.set_window(w2.x, w2.y, w2.t_ns, 2, w2.knots_init, w2.start_ns, w2.dt_ns, w2.num_fixed, w2.t_next_win_beg_ns,
                  IG="resident")
    IGh = ref.IG.copy()
    ref.set_window(w2.x, w2.y, w2.t_ns, w2.knots_init, w2.start_ns, w2.dt_ns, w2.num_fixed, w2.t_next_win_beg_ns, IGh)
    for d in (np.zeros(w2.P), np.full(w2.P, -0.003)):
        c, g = be.eval(d)
        c_ref, g_ref = ref.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
    assert ref.alpha > 0 and rel_scalar(be.alpha, ref.alpha) < RTOL

    be.resetIG()
    IG, visits = be.getIG(with_visits=True)
    assert IG.max() == 0 and visits.max() == 0


def test_set_and_get_map_roundtrip(hip):
    w = synth.backend_window(2_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 256, 128, 2, 5, 1, 0.2, seed=3)
    be = hip.reference_shaped.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    rng = np.random.default_rng(0)
    IG = rng.random((w.Hp, w.Wp)).astype(np.float32)
    v = rng.integers(0, 255, (w.Hp, w.Wp)).astype(np.uint8)
    be.setIG(IG, v)
    a, b = be.getIG(with_visits=True)
    np.testing.assert_array_equal(a, IG)
    np.testing.assert_array_equal(b, v)
    with pytest.raises(hip.CmaxHipError):
        be.updateIG(3)  # no evaluation has run: IL_old undefined
