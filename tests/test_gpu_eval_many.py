"""-m gpu: cmx_*_eval_many -- m independent evaluations queued back to back, one wait -- returns what m single evaluations
return (and what the oracle returns), for both ends, with and without gradients, and leaves the context usable."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fast", [True, False])
def test_frontend_eval_many(hip, oracle, fast):
    p = synth.frontend_packet(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=81)
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    if fast:
        fe.set_fast_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    rng = np.random.default_rng(5)
    xs = np.vstack([np.zeros(3), p.omega_true] + [p.omega_true * rng.uniform(0, 1.3) + rng.normal(0, 0.2, 3) for _ in range(7)])
    c, g = fe.eval_many(xs, True)
    c0, _ = fe.eval_many(xs, False)
    refs = [ref.eval(x) for x in xs]
    # xs[1] is the packet's true rate: the optimum, where the gradient is a small difference of large sums.  The reference-shaped path
    # adds its votes with unordered fp32 global atomics: two runs of the SAME evaluation differ by ~4e-6 of that small gradient (and
    # once in ten runs of the whole suite by more than 1e-5: a flaky failure of this test, round 6).  Its gradients are therefore
    # compared on the gradient scale of the problem (the largest |gradient| over the test's points), which is what the rounding of
    # the accumulators is relative to; the production path (LDS fixed-point windows) keeps the strict per-point bound.
    gscale = max(float(np.abs(gr).max()) for _, gr in refs)

    def grad_err(a, b):
        return rel_vec(a, b) if fast else float(np.abs(np.asarray(a) - np.asarray(b)).max()) / gscale
    for i, x in enumerate(xs):
        c_ref, g_ref = refs[i]
        assert rel_scalar(c[i], c_ref) < RTOL and rel_scalar(c0[i], c_ref) < RTOL, i
        assert grad_err(g[i], g_ref) < RTOL, (i, g[i], g_ref)
        cs, gs = fe.eval(x)          # the context keeps working, and agrees with its own single evaluations
        assert rel_scalar(c[i], cs) < (1e-6 if fast else RTOL) and grad_err(g[i], gs) < RTOL, (i, c[i], cs, g[i], gs)
    assert fe.eval_many(np.zeros((0, 3)), True)[0].size == 0
    x, rep = fe.setupProblemAndOptimize(np.zeros(3))
    assert rep["final_cost"] < rep["initial_cost"]


def test_backend_eval_many(hip, oracle):
    w = synth.backend_window(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=82)
    IG = np.zeros((w.Hp, w.Wp), np.float32)
    IG[100:140, 200:300] = 1.1
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE, IG)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, IG)
    rng = np.random.default_rng(6)
    xs = np.vstack([np.zeros(w.P)] + [rng.normal(0, 0.01, w.P) for _ in range(5)])   # first point = 0: alpha is fixed there
    c, g = be.eval_many(xs, True)
    c0, _ = be.eval_many(xs, False)
    for i, x in enumerate(xs):
        c_ref, g_ref = ref.eval(x)
        assert rel_scalar(c[i], c_ref) < RTOL and rel_scalar(c0[i], c_ref) < RTOL, i
        assert rel_vec(g[i], g_ref) < RTOL, i
    assert rel_scalar(be.alpha, ref.alpha) < RTOL and ref.alpha > 0
    with pytest.raises(hip.CmaxHipError):
        be.comm_attach(be.comm_unique_id(), 0, 1)
        be.eval_many(xs, True)       # not with a communicator attached


def test_eval_each_is_a_sequence_of_single_evaluations(hip):
    """cmx_*_eval_each = m calls of cmx_*_eval inside one native call: same bits in deterministic mode, same reuse rules."""
    p = synth.config1()
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_option(_lib.OPT_DETERMINISTIC, 1)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    xs = np.array([[0.0, 0.0, 0.0], [0.1, -0.2, 0.05], [0.1, -0.2, 0.05], [0.6, -0.9, 0.4]])
    single = [fe.eval(x) for x in xs]
    c, g = fe.eval_each(xs, True)
    c0, g0 = fe.eval_each(xs, False)
    for i, (cs, gs) in enumerate(single):
        assert c[i] == cs and np.array_equal(g[i], gs) and c0[i] == cs, i
    assert g0 is None and fe.eval_each(np.zeros((0, 3)))[0].size == 0
    w = synth.backend_window(20_000, 120, 90, 130.0, 130.0, 59.5, 44.5, 512, 256, 4, 8, 2, 0.2, seed=9)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_option(_lib.OPT_DETERMINISTIC, 1)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE)
    rng = np.random.default_rng(3)
    xb = np.vstack([np.zeros(w.P)] + [rng.normal(0, 0.01, w.P) for _ in range(3)])
    single = [be.eval(x) for x in xb]
    c, g = be.eval_each(xb, True)
    for i, (cs, gs) in enumerate(single):
        assert c[i] == cs and np.array_equal(g[i], gs), i
