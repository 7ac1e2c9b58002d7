"""-m gpu: the back end's tile-ordered gradient pass (CMX_OPT_TILE_GATHER, default): the splat leaves (vote cell, dx, dy) per
tile-ordered event, the gradient pass walks the same order and applies the batch's spline Jacobian per event
(event_pano_warper.cpp:277-332 is the loop both restate).  Against the CPU oracle and against the time-ordered passes it
replaces, over the cases that take different code paths: both spline orders, fixed knots (the j >= 0 column rule), sampling,
votes next to the panorama border (the mu term), a non-zero global map, f-then-df on the resident image, the gated pass of
a solve, independent evaluations queued back to back, a jump of the parameters that re-sorts the events."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _pair(hip, oracle, w, rate=1, sigma=1.0, IG=None, tile=True):
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_option(_lib.OPT_TILE_GATHER, 1 if tile else 0)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, rate,
                  sigma, _lib.VARIANCE, IG)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, rate, sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    return be, ref


@pytest.mark.parametrize("order,K,nf,T,rate", [(2, 5, 0, 0.2, 1), (2, 6, 1, 0.25, 3), (4, 10, 3, 0.35, 1), (4, 7, 0, 0.2, 2), (4, 12, 5, 0.4, 1)])
def test_tile_gather_matches_oracle_and_time_ordered_passes(hip, oracle, order, K, nf, T, rate):
    w = synth.backend_window(70_001, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, order, K, nf, T, seed=7 + K)
    be, ref = _pair(hip, oracle, w, rate=rate)
    old, _ = _pair(hip, oracle, w, rate=rate, tile=False)
    rng = np.random.default_rng(K)
    for scale in (0.0, 0.004, 0.03):
        d = rng.normal(0, scale, w.P) if scale else np.zeros(w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        c0, g0 = old.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (scale, rel_vec(g, g_ref))
        assert rel_vec(g0, g_ref) < RTOL and rel_vec(g, g0) < RTOL
    assert be.stats()["tile_evals"] == 3 and old.stats()["tile_evals"] == 0


def test_tile_gather_border_votes_and_global_map(hip, oracle):
    """A panorama so small that the camera's band reaches its top and bottom rows: votes within r of the border carry the mu term
    (S2); with a non-zero global map alpha != 0."""
    w = synth.backend_window(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 200, 100, 4, 8, 2, 0.25, seed=23)
    yy, xx = np.mgrid[0:w.Hp, 0:w.Wp]
    IG = (2.0 * np.exp(-((xx - 100) ** 2 + (yy - 50) ** 2) / 200.0)).astype(np.float32)
    for sigma in (1.0, 2.0):
        be, ref = _pair(hip, oracle, w, sigma=sigma, IG=IG)
        d = np.random.default_rng(2).normal(0, 0.01, w.P)
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert ref.alpha > 0 and rel_scalar(be.alpha, ref.alpha) < RTOL
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, sigma
        assert be.stats()["tile_evals"] == 1


def test_tile_gather_f_then_df_gated_solve_and_eval_many(hip, oracle):
    w = synth.backend_window(90_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=31)
    be, ref = _pair(hip, oracle, w)
    d = np.random.default_rng(4).normal(0, 0.004, w.P)
    c_ref, g_ref = ref.eval(d)
    # f, then df at the same point: the gradient pass consumes the records the cost-only evaluation's splat left
    c0, _ = be.eval(d, False)
    c, g = be.eval(d, True)
    assert be.stats()["reuse_hits"] == 1 and be.stats()["tile_evals"] == 1
    assert rel_scalar(c0, c_ref) < RTOL and rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
    # a jump that moves every vote out of its window: exact through the global path, re-sorted at the next evaluation
    big = np.full(w.P, 0.3)
    cb_ref, gb_ref = ref.eval(big)
    cb, gb = be.eval(big)
    assert rel_scalar(cb, cb_ref) < RTOL and rel_vec(gb, gb_ref) < RTOL
    big2 = big * 1.0001
    cb_ref, gb_ref = ref.eval(big2)
    cb, gb = be.eval(big2)
    assert rel_scalar(cb, cb_ref) < RTOL and rel_vec(gb, gb_ref) < RTOL
    assert be.stats()["rebins"] >= 2
    # independent evaluations queued back to back: every one's gradient pass reads the records of ITS splat
    xs = np.stack([np.random.default_rng(10 + k).normal(0, 0.004, w.P) for k in range(5)])
    cs, gs = be.eval_many(xs, True)
    for k in range(5):
        ck_ref, gk_ref = ref.eval(xs[k])
        assert rel_scalar(cs[k], ck_ref) < RTOL and rel_vec(gs[k], gk_ref) < RTOL, k
    # the solve (gated gradient passes behind the line search's cost-only probes) lands where the time-ordered passes land
    old, _ = _pair(hip, oracle, w, tile=False)
    x1, r1 = be.setupProblemAndOptimize()
    x0, r0 = old.setupProblemAndOptimize()
    assert r1["final_cost"] < r1["initial_cost"] and abs(r1["final_cost"] - r0["final_cost"]) < 2e-3 * abs(r0["final_cost"])
    assert be.stats()["gated_hits"] > 0


def test_tile_gather_falls_back_where_it_cannot_run(hip, oracle):
    """Deterministic mode and a bearing table with z != 1 (no per-event streams) keep the time-ordered passes; results are right."""
    w = synth.backend_window(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 256, 128, 2, 5, 1, 0.2, seed=16)
    d = np.full(w.P, 0.003)
    be, ref = _pair(hip, oracle, w)
    be.set_deterministic(True)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL and be.stats()["tile_evals"] == 0
    lut = w.lut.copy().reshape(-1, 3)
    lut *= np.linspace(0.9, 1.1, len(lut))[:, None]   # same rays, z != 1: the general three-component table
    be2 = hip.BackendEvaluator(w.W, w.H, lut.reshape(w.lut.shape), w.Wp, w.Hp)
    be2.set_fast_path()
    be2.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, 1, 1.0,
                   _lib.VARIANCE)
    ref2 = oracle.Backend(w.W, w.H, lut.reshape(w.lut.shape), w.Wp, w.Hp, w.order)
    ref2.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    c2_ref, g2_ref = ref2.eval(d)
    c2, g2 = be2.eval(d)
    assert rel_scalar(c2, c2_ref) < RTOL and rel_vec(g2, g2_ref) < RTOL and be2.stats()["tile_evals"] == 0
