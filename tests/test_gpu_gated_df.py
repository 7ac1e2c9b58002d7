"""-m gpu: cmx_hint_next_df / CMX_OPT_GATED_DF -- the gradient pass queued behind a cost-only evaluation and gated on the device
by that evaluation's own cost.  Whatever the hint says, results are those of the plain call sequence; the hint only decides
whether the df call finds its result in flight (stats: gated_launches / gated_hits)."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _fe(hip, det=False):
    p = synth.config1()
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    if det:
        fe.set_option(_lib.OPT_DETERMINISTIC, 1)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    return p, fe


def _be(hip, batch=100):
    w = synth.backend_window(40_000, 160, 120, 190.0, 190.0, 79.5, 59.5, 512, 256, 4, 8, 2, 0.25, seed=12)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, batch, 1, 1.0,
                  _lib.VARIANCE)
    return w, be


@pytest.mark.parametrize("kind", ["fe", "fe_det", "be", "be_unfolded"])
def test_hinted_sequences_return_the_plain_results(hip, oracle, kind):
    if kind.startswith("fe"):
        p, ev = _fe(hip, det=(kind == "fe_det"))
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        pts = [np.array(v) for v in ([0.0, 0.0, 0.0], [0.1, -0.2, 0.05], [0.5, -0.8, 0.3], [0.6, -0.9, 0.4])]
    else:
        w, ev = _be(hip, batch=(100 if kind == "be" else 50))     # batch 50: no four-events-per-lane pass, nothing to gate
        ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, 100 if kind == "be" else 50, 1, 1.0, _lib.VARIANCE)
        ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, None)
        rng = np.random.default_rng(2)
        pts = [np.zeros(w.P)] + [rng.normal(0, 0.01, w.P) for _ in range(3)]
    expected = [ref.eval(x) for x in pts]
    gates = kind != "be_unfolded"

    def check(i, c, g):
        c_ref, g_ref = expected[i]
        assert rel_scalar(c, c_ref) < RTOL, (kind, i)
        if g is not None:
            assert rel_vec(g, g_ref) < RTOL, (kind, i)

    s0 = ev.stats()
    # 1. gate open (mode 4 = always): the df is served by the pass in flight
    ev.hint_next_df(0.0, 4)
    check(1, *ev.eval(pts[1], False))
    check(1, *ev.eval(pts[1], True))
    s1 = ev.stats()
    assert s1["gated_launches"] - s0["gated_launches"] == (1 if gates else 0)
    assert s1["gated_hits"] - s0["gated_hits"] == (1 if gates else 0)
    # 2. the test on the cost itself: f = -contrast < threshold, threshold just above / just below the value
    f1 = -expected[2][0]
    for thr, opens in ((f1 + abs(f1) * 1e-3, True), (f1 - abs(f1) * 1e-3, False)):
        before = ev.stats()["gated_hits"]
        ev.hint_next_df(thr, 1)
        check(2, *ev.eval(pts[2], False))
        check(2, *ev.eval(pts[2], True))          # asked for either way: right either way
        assert ev.stats()["gated_hits"] - before == (1 if (opens and gates) else 0), (kind, thr, opens)
        check(2, *ev.eval(pts[2], True))          # a second df at the same point: ordinary pass on the resident image
    # 3. gate open but nobody asks: the next evaluation is elsewhere
    ev.hint_next_df(0.0, 4)
    check(3, *ev.eval(pts[3], False))
    check(0, *ev.eval(pts[0], False))
    check(0, *ev.eval(pts[0], True))
    check(3, *ev.eval(pts[3], True))
    # 4. a hint is consumed by ONE evaluation; a withdrawn hint queues nothing
    n = ev.stats()["gated_launches"]
    check(1, *ev.eval(pts[1], False))
    ev.hint_next_df(0.0, 4)
    ev.hint_next_df(0.0, 0)
    check(2, *ev.eval(pts[2], False))
    assert ev.stats()["gated_launches"] == n
    # 5. NaN threshold with mode 3 (not f >= thr) opens the gate, like the C expression does
    ev.hint_next_df(float("nan"), 3)
    check(1, *ev.eval(pts[1], False))
    before = ev.stats()["gated_hits"]
    check(1, *ev.eval(pts[1], True))
    assert ev.stats()["gated_hits"] - before == (1 if gates else 0)
    # 6. option off: hints are ignored
    ev.set_option(_lib.OPT_GATED_DF, 0)
    n = ev.stats()["gated_launches"]
    ev.hint_next_df(0.0, 4)
    check(2, *ev.eval(pts[2], False))
    check(2, *ev.eval(pts[2], True))
    assert ev.stats()["gated_launches"] == n


@pytest.mark.parametrize("kind", ["fe", "be"])
def test_solver_takes_the_same_path_with_and_without_the_gated_pass(hip, kind):
    reps = []
    for gated in (0, 1):
        if kind == "fe":
            p, ev = _fe(hip, det=True)        # deterministic: the two runs must agree to the bit
            x0 = np.zeros(3)
        else:
            w, ev = _be(hip)
            ev.set_option(_lib.OPT_DETERMINISTIC, 0)
            x0 = np.zeros(w.P)
        ev.set_option(_lib.OPT_GATED_DF, gated)
        x, rep = ev.setupProblemAndOptimize(x0.copy())
        st = ev.stats()
        reps.append((np.asarray(x), rep, st))
    (xa, ra, sa), (xb, rb, sb) = reps
    assert sa["gated_launches"] == 0 and sb["gated_launches"] >= rb["iterations"] and sb["gated_hits"] >= rb["iterations"] - 1
    if kind == "fe":   # deterministic: the same evaluations, the same bits
        assert (ra["iterations"], ra["n_f"], ra["n_df"], ra["status"]) == (rb["iterations"], rb["n_f"], rb["n_df"], rb["status"])
        assert np.array_equal(xa, xb) and ra["final_cost"] == rb["final_cost"]
    else:              # fp32 atomics: two runs of the same solve may branch differently at a near-tie of the line search
        assert ra["status"] == rb["status"] and abs(ra["final_cost"] - rb["final_cost"]) < 1e-4 * abs(ra["final_cost"])
        assert np.abs(xa - xb).max() < 5e-3
