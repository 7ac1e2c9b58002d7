"""-m gpu: CMX_OPT_FUSED_GATHER (opt-in) -- the front-end gradient pass with the image pass fused in: Jt = G^T G I built per
chunk window in LDS through the banded composite operator, image moments from the event-side sums.  Against the CPU
oracle, and against the default flow, on images whose border tiles dominate, for every radius the kernel is instantiated
for, with drifted parameters (votes leaving their windows take the direct-operator path)."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _fe(hip, p, sigma, measure, fused):
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_option(_lib.OPT_FUSED_GATHER, 1 if fused else 0)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
    return fe


@pytest.mark.parametrize("sigma", [0.5, 0.8, 1.0])       # radius 2, 3, 4
@pytest.mark.parametrize("W,H", [(240, 180), (70, 50), (333, 97)])
def test_fused_gather_matches_the_oracle(hip, oracle, W, H, sigma):
    f = 0.9 * max(W, H)
    p = synth.frontend_packet(40_000, W, H, f, f, (W - 1) / 2, (H - 1) / 2, seed=61)
    for measure in (_lib.VARIANCE, _lib.MEAN_SQUARE):
        fe = _fe(hip, p, sigma, measure, True)
        ref = oracle.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, sigma, measure)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        # the tile sort is taken at the first point; the later ones drift away from it (small, then large: > 2 % of the
        # votes outside their windows makes the evaluation fall back to the separate image pass -- results stay right)
        for om in ([0.3, -0.5, 0.2], [0.35, -0.55, 0.25], [0.3, -0.5, 0.2], p.omega_true, [-2.0, 1.5, 3.0], [0.3, -0.5, 0.2]):
            c_ref, g_ref = ref.eval(om)
            c, g = fe.eval(om)
            assert rel_scalar(c, c_ref) < RTOL, (W, H, sigma, measure, om, c, c_ref)
            assert rel_vec(g, g_ref) < RTOL, (W, H, sigma, measure, om, g, g_ref)
            assert rel_scalar(fe.eval(om, False)[0], c_ref) < RTOL
        assert fe.stats()["fused_evals"] >= 2


def test_fused_gather_equals_the_default_flow_and_is_used_by_the_solver(hip):
    p = synth.config1()
    a, b = _fe(hip, p, p.sigma, _lib.VARIANCE, True), _fe(hip, p, p.sigma, _lib.VARIANCE, False)
    for om in ([0.0, 0.0, 0.0], [0.1, -0.2, 0.05], [0.1, -0.2, 0.05]):
        ca, ga = a.eval(om)
        cb, gb = b.eval(om)
        assert rel_scalar(ca, cb) < 1e-6 and rel_vec(ga, gb) < 1e-6
    assert a.stats()["fused_evals"] >= 2 and b.stats()["fused_evals"] == 0
    xa, ra = a.setupProblemAndOptimize(np.zeros(3))
    xb, rb = b.setupProblemAndOptimize(np.zeros(3))
    assert abs(ra["final_cost"] - rb["final_cost"]) < 2e-3 * abs(rb["final_cost"]) and np.abs(xa - xb).max() < 0.05
    # sigma whose radius the kernel is not instantiated for: the option is ignored
    c = _fe(hip, p, 2.0, _lib.VARIANCE, True)
    c.eval([0.1, -0.2, 0.05])
    assert c.stats()["fused_evals"] == 0
