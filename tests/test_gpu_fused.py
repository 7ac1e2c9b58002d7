"""-m gpu: tile-dataflow fusion of the adjoint image pass into the front-end splat (CMX_OPT_FUSED_IMAGE, round 6).

A gradient evaluation of the production path is two launches (splat + image pass, gather) instead of three.  The fused form
must give the separate launches' numbers (same per-pixel arithmetic; only the grouping of the fp64 moment sums differs) and the
oracle's (local_image_warped_events.cpp:10-170, local_focus_funcs.cpp:26-44) within north_star's 1e-5, for any parameters --
including jumps that throw votes out of their LDS windows (the evaluation is then repeated through the separate launches)."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _fe(hip, p, fused, measure=0, sigma=None):
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_option(_lib.OPT_FUSED_IMAGE, int(fused))  # 0: three launches, 1 (default): two, 2: one (A/B form)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma if sigma is None else sigma, measure)
    return fe


@pytest.fixture(scope="module")
def small():
    return synth.frontend_packet(60_013, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=21)


@pytest.mark.parametrize("measure", [0, 1])
def test_fused_equals_separate_launches_and_oracle(hip, oracle, small, measure):
    p = small
    a, b = _fe(hip, p, True, measure), _fe(hip, p, False, measure)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, measure)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    rng = np.random.default_rng(5)
    om = np.array([0.55, -0.85, 0.35])
    for it in range(12):
        ca, ga = a.eval(om)
        cb, gb = b.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(ca, cb) < 1e-7 and rel_vec(ga, gb) < 1e-6, (it, ca, cb, ga, gb)
        assert rel_scalar(ca, cr) < RTOL and rel_vec(ga, gr) < RTOL, (it, ca, cr, ga, gr)
        om = om + rng.normal(0, 0.03, 3)  # a line search's steps: a pixel or two
    sa, sb = a.stats(), b.stats()
    assert sa["fused_evals"] == 12 and sa["fused_redos"] == 0, sa
    assert sb["fused_evals"] == 0, sb


def test_cost_only_and_reuse_around_fused_evaluations(hip, oracle, small):
    """f, then df at the same point (conjugate_fr's pattern), then fdf elsewhere: every mix of the fused gradient
    evaluation with the cost-only path and the image reuse gives the oracle's numbers."""
    p = small
    fe = _fe(hip, p, True)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    pts = [(0.5, -0.8, 0.3), (0.52, -0.83, 0.31), (0.6, -0.9, 0.4)]
    for om in pts:
        cr, gr = ref.eval(om)
        f, df = fe.contrast_fdf(om)                               # a new point: fused
        assert rel_scalar(-f, cr) < RTOL and rel_vec(-df, gr) < RTOL
        f2, df2 = fe.contrast_fdf(om)                             # the same point again: the resident image is reused
        assert rel_scalar(-f2, cr) < RTOL and rel_vec(-df2, gr) < RTOL
        om2 = (om[0] + 0.01, om[1], om[2] - 0.01)
        cr2, gr2 = ref.eval(om2)
        assert rel_scalar(-fe.contrast_f(om2), cr2) < RTOL        # cost-only (separate launches, Jt kept)
        assert rel_vec(-fe.contrast_df(om2), gr2) < RTOL          # served from the resident image / the gated pass
    s = fe.stats()
    assert s["fused_evals"] == 3 and s["fused_redos"] == 0, s


def test_jump_out_of_the_windows_is_repeated_and_resorted(hip, oracle, small):
    """A jump of omega far beyond the 16-pixel window margin: the fused evaluation reports votes on the global path, is
    repeated through the separate launches (exact for any parameters) and the events are sorted again."""
    p = small
    fe = _fe(hip, p, True)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    seq = [(0.0, 0.0, 0.0), (6.0, -5.0, 9.0), (6.0, -5.0, 9.0), (6.02, -5.0, 9.0), (-4.0, 3.0, -8.0), (0.0, 0.0, 0.0)]
    for om in seq:
        c, g = fe.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL, (om, c, cr, g, gr)
    s = fe.stats()
    assert s["fused_redos"] >= 2 and s["rebins"] >= 3, s


def test_moves_beyond_the_window_but_within_reach_need_no_repeat(hip, oracle, small):
    """A line search's range (omega from 0 to the packet's true rate and back: up to ~18 px of motion): votes leave their 16-px LDS
    margin and take the global path, but stay within the 56 px the tiles' arrival counts cover -- no evaluation is repeated."""
    p = small
    fe = _fe(hip, p, True)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for s_ in (0.0, 1.0, 0.3, 2.5, 0.0, -1.5):
        om = np.array([0.6, -0.9, 0.4]) * s_ * 3.0
        c, g = fe.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL, (om, c, cr, g, gr)
    s = fe.stats()
    assert s["fused_evals"] == 6 and s["fused_redos"] == 0, s


@pytest.mark.parametrize("W,H", [(64, 48), (100, 70), (346, 260), (33, 35)])
def test_partial_tiles_and_small_images(hip, oracle, W, H):
    """Image sides that are not multiples of the 32-pixel tile, images of a few tiles: border folding of the banded operator,
    reflected halos and partially filled tiles."""
    p = synth.frontend_packet(20_011, W, H, 0.9 * W, 0.9 * W, (W - 1) / 2, (H - 1) / 2, seed=W + H)
    a, b = _fe(hip, p, True), _fe(hip, p, False)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for om in [(0.2, -0.4, 0.3), (0.25, -0.42, 0.28), (-0.6, 0.5, 1.0)]:
        ca, ga = a.eval(om)
        cb, gb = b.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(ca, cb) < 1e-7 and rel_vec(ga, gb) < 1e-6
        assert rel_scalar(ca, cr) < RTOL and rel_vec(ga, gr) < RTOL


def test_other_blur_radii_keep_the_separate_launches(hip, oracle, small):
    p = small
    for sigma in (0.0, 0.5, 2.0):
        fe = _fe(hip, p, True, sigma=sigma)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, 0)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        c, g = fe.eval((0.4, 0.1, -0.7))
        cr, gr = ref.eval((0.4, 0.1, -0.7))
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL
        assert fe.stats()["fused_evals"] == 0


def test_config2_full_size_fused(hip, oracle):
    """BASELINE config 2 (1M events, 640x480) through the two-launch evaluation, at a sequence of points."""
    p = synth.frontend_packet(1_000_000, 640, 480, 588.10, 593.99, 339.83, 242.43, seed=20240316)
    fe = _fe(hip, p, True)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for om in [(0.6, -0.9, 0.4), (0.55, -0.95, 0.42), (0.62, -0.88, 0.37)]:
        c, g = fe.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL, (om, c, cr, g, gr)
    s = fe.stats()
    assert s["fused_evals"] == 3 and s["fused_redos"] == 0, s


def test_many_fused_evaluations_leave_the_counters_clean(hip, small):
    """400 evaluations back to back on one context, two more contexts interleaved: the arrival counters reset themselves, the
    ping-pong partner is cleared by the tiles' passes, nothing drifts."""
    p = small
    fe, other = _fe(hip, p, True), _fe(hip, p, True)
    plain = _fe(hip, p, False)
    rng = np.random.default_rng(9)
    om = np.array([0.5, -0.8, 0.3])
    for it in range(400):
        c, g = fe.eval(om)
        if it % 50 == 0:
            cb, gb = plain.eval(om)
            assert rel_scalar(c, cb) < 1e-7 and rel_vec(g, gb) < 1e-6, it
            other.eval(om + 0.01)
        om = om + rng.normal(0, 0.01, 3)
    s = fe.stats()
    assert s["fused_evals"] == 400 and s["fused_redos"] == 0 and s["rebins"] == 1, s


@pytest.mark.parametrize("measure", [0, 1])
def test_one_launch_form_matches_too(hip, oracle, small, measure):
    """CMX_OPT_FUSED_IMAGE = 2 (kept as an A/B switch: measured slower): gather and finalize inside the splat launch as well.  Same
    numbers as the other two forms and the oracle's, incl. after a jump beyond the tiles' reach and on a partial-tile image."""
    for p in (small, synth.frontend_packet(20_011, 100, 70, 90.0, 90.0, 49.5, 34.5, seed=170)):
        a, b = _fe(hip, p, 2, measure), _fe(hip, p, 0, measure)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, measure)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        for om in [(0.5, -0.8, 0.3), (0.52, -0.83, 0.31), (6.0, -5.0, 9.0), (6.0, -5.01, 9.0), (0.0, 0.0, 0.0)]:
            ca, ga = a.eval(om)
            cb, gb = b.eval(om)
            cr, gr = ref.eval(om)
            assert rel_scalar(ca, cb) < 1e-7 and rel_vec(ga, gb) < 1e-6, (om, ca, cb, ga, gb)
            assert rel_scalar(ca, cr) < RTOL and rel_vec(ga, gr) < RTOL, (om, ca, cr, ga, gr)
        s = a.stats()
        assert s["one_launch_evals"] >= 5, s
        assert rel_scalar(-a.contrast_f((0.1, 0.2, 0.3)), ref.eval((0.1, 0.2, 0.3))[0]) < RTOL  # the other paths still work behind it


@pytest.mark.parametrize("measure", [0, 1])
def test_self_service_one_launch_form(hip, oracle, small, measure):
    """CMX_OPT_FUSED_IMAGE = 3 (an A/B switch: in its correct form -- an agent-scope acquire behind the tiles' stamps -- slower than the
    default, profiles/r06_selfserve.txt): ONE launch of the chunk workgroups alone -- each runs the image pass of the tiles it owns and
    gathers the gradient sums of its own events (cmx_selfserve.hpp).  Same numbers as the three-launch form and the oracle's, on a dense
    packet, a partial-tile image and a SPARSE packet (fewer chunks than tiles: workgroups own several tiles), incl. a jump beyond the
    tiles' reach (repeated after a fresh sort) and motion onto the global path (waits for every pass of the launch)."""
    sparse = synth.frontend_packet(3_001, 346, 260, 300.0, 300.0, 172.5, 129.5, seed=33)
    for p in (small, synth.frontend_packet(20_011, 100, 70, 90.0, 90.0, 49.5, 34.5, seed=170), sparse):
        a, b = _fe(hip, p, 3, measure), _fe(hip, p, 0, measure)
        ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, measure)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        pts = [(0.5, -0.8, 0.3), (0.52, -0.83, 0.31), (0.53, -0.80, 0.33), (1.6, -2.4, 1.1), (1.62, -2.4, 1.1), (6.0, -5.0, 9.0), (6.0, -5.01, 9.0),
               (6.01, -5.01, 9.0), (0.0, 0.0, 0.0), (0.01, 0.0, 0.0)]
        for om in pts:
            ca, ga = a.eval(om)
            cb, gb = b.eval(om)
            cr, gr = ref.eval(om)
            assert rel_scalar(ca, cb) < 1e-7 and rel_vec(ga, gb) < 1e-6, (om, ca, cb, ga, gb)
            assert rel_scalar(ca, cr) < RTOL and rel_vec(ga, gr) < RTOL, (om, ca, cr, ga, gr)
        s = a.stats()
        assert s["self_serve_evals"] >= 4 and s["fused_timeouts"] == 0, s  # (a sort's first evaluation does not know the table's length yet)
        assert rel_scalar(-a.contrast_f((0.1, 0.2, 0.3)), ref.eval((0.1, 0.2, 0.3))[0]) < RTOL  # the other paths still work behind it


def test_self_service_config2_full_size_and_many_evaluations(hip, oracle):
    """BASELINE config 2 (1M events, 640x480) through the self-service launch at a sequence of points, then 300 evaluations back to
    back against the three-launch form: counters, stamps and accumulator rows reset themselves."""
    p = synth.frontend_packet(1_000_000, 640, 480, 588.10, 593.99, 339.83, 242.43, seed=20240316)
    fe, plain = _fe(hip, p, 3), _fe(hip, p, 0)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for om in [(0.6, -0.9, 0.4), (0.55, -0.95, 0.42), (0.62, -0.88, 0.37)]:
        c, g = fe.eval(om)
        cr, gr = ref.eval(om)
        assert rel_scalar(c, cr) < RTOL and rel_vec(g, gr) < RTOL, (om, c, cr, g, gr)
    rng = np.random.default_rng(19)
    om = np.array([0.6, -0.9, 0.4])
    for it in range(300):
        c, g = fe.eval(om)
        if it % 30 == 0:
            cb, gb = plain.eval(om)
            assert rel_scalar(c, cb) < 1e-7 and rel_vec(g, gb) < 1e-6, (it, c, cb, g, gb)
        om = om + rng.normal(0, 0.01, 3)
    s = fe.stats()
    assert s["self_serve_evals"] >= 300 and s["fused_redos"] == 0 and s["fused_timeouts"] == 0, s


def test_self_service_beside_other_contexts_stays_correct(hip):
    """Three host threads, each with its own self-service context on the same GPU: a launch may now find CUs taken by another context's
    workgroups, i.e. NOT all of its workgroups resident at once -- the case the form's bounded waits exist for.  Whatever happens
    (no wait runs out; or some do, the evaluation is repeated through the separate launches and after three strikes the context
    stops using the form), every evaluation returns the three-launch form's numbers and nothing hangs.  This is the test that found
    (i) the repeat path trusting a ping-pong partner that passes which gave up never cleared and (ii) the form's cached loads of Jt
    returning stale lines of the previous evaluation about once in 2000 evaluations -- only under sharing."""
    import threading
    p = synth.frontend_packet(300_007, 640, 480, 588.10, 593.99, 339.83, 242.43, seed=77)
    plain = _fe(hip, p, 0)
    rng = np.random.default_rng(3)
    pts = [np.array([0.6, -0.9, 0.4]) + rng.normal(0, 0.02, 3) for _ in range(40)]
    want = [plain.eval(om) for om in pts]
    errs, stats = [], []

    def work(k):
        try:
            fe = _fe(hip, p, 3)
            for rep in range(5):
                for om, (cw, gw) in zip(pts, want):
                    c, g = fe.eval(om)
                    if not (rel_scalar(c, cw) < 1e-7 and rel_vec(g, gw) < 1e-6):
                        st = fe.stats()
                        errs.append((k, rep, c, cw, list(g), list(gw), rel_vec(g, gw), st["self_serve_evals"], st["fused_timeouts"], st["fused_redos"]))
                        return
            stats.append(fe.stats())
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a self-service context hangs beside other contexts"
    assert not errs, errs[:3]
    assert len(stats) == 3 and all(s["fused_evals"] >= 200 for s in stats), stats
    print("self-service beside other contexts:", [(s["self_serve_evals"], s["fused_timeouts"], s["fused_redos"]) for s in stats])


def test_take_overs_in_mid_launch_short_soak():
    """tools/soak_fused.py for a few seconds: six host threads, each checking a fused context against a three-launch one evaluation by
    evaluation, with device-driven solves the host takes over in mid-launch (the stop word lands while a slot's workgroups are running),
    jumps beyond the tiles' reach and a back-end context beside them.  Round 6's long runs of this found the stop word being read per
    wave (profiles/r06_soak_fused.txt); here: no mismatch, no tile that gave up waiting."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_fused.py"), "10", "6"], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "soak ok" in out, out[-3000:]
    assert "TIMEOUTS 0;" in out, out[-3000:]
    assert "chain take-overs 0," not in out, out[-3000:]   # the case under test did occur
