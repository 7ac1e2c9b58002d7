"""-m gpu: the FR-CG driver loops (cmx_frontend_solve / cmx_backend_solve, host C++) over the HIP evaluator, against
the SAME driver run over the CPU oracle's cost functor (cmax_slam_amd.solver.frcg_minimize)."""
import numpy as np
import pytest

from cmax_slam_amd import solver, synth

pytestmark = pytest.mark.gpu


def _oracle_fdf(ref):
    def fdf(x, wg):
        c, g = ref.eval(x, wg)
        return -c, (-g if wg else None)
    return fdf


@pytest.mark.parametrize("fast", [False, True])
def test_frontend_solve_matches_the_oracle_solve(hip, oracle, fast):
    p = synth.frontend_packet(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=33)
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    if fast:
        fe.set_fast_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    x_ref, rep_ref = solver.frcg_minimize(_oracle_fdf(ref), np.zeros(3), **solver.FRONTEND)
    x, rep = fe.setupProblemAndOptimize(np.zeros(3))
    # identical optimiser, costs equal to ~1e-8: the iterates coincide until rounding decides a stopping test
    assert abs(rep["final_cost"] - rep_ref["final_cost"]) < 1e-3 * abs(rep_ref["final_cost"])
    assert np.abs(x - x_ref).max() < 0.02
    assert abs(rep["iterations"] - rep_ref["iterations"]) <= 2
    assert rep["initial_cost"] == pytest.approx(rep_ref["initial_cost"], rel=1e-6)
    assert np.abs(x[:2] - p.omega_true[:2]).max() < 0.05
    if fast:
        assert fe.stats()["chain_solves"] == 1 and fe.stats()["chain_takeovers"] == 0   # the line search ran ahead on the device
        from cmax_slam_amd import _lib
        fe.set_option(_lib.OPT_CHAIN_SOLVE, 0)                                         # ... and host-driven: the same solve
        x2, rep2 = fe.setupProblemAndOptimize(np.zeros(3))
        assert abs(rep2["final_cost"] - rep_ref["final_cost"]) < 1e-3 * abs(rep_ref["final_cost"]) and np.abs(x2 - x_ref).max() < 0.02
        assert fe.stats()["reuse_hits"] >= rep2["iterations"] - 1  # df after f at accepted points reused the image


def test_frontend_solve_warm_start_never_worsens(hip):
    p = synth.frontend_packet(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=33)
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    x0, rep0 = fe.setupProblemAndOptimize(np.zeros(3))
    x1, rep1 = fe.setupProblemAndOptimize(x0)  # the reference keeps ang_vel_ between packets
    assert rep1["initial_cost"] == pytest.approx(rep0["final_cost"], rel=1e-6)
    assert rep1["final_cost"] <= rep0["final_cost"] + 1e-9


@pytest.mark.parametrize("order,K,nf,T", [(2, 5, 1, 0.2), (4, 10, 3, 0.35)])
def test_backend_solve_matches_the_oracle_solve(hip, oracle, order, K, nf, T):
    w = synth.backend_window(40_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, order, K, nf, T, seed=5)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    x_ref, rep_ref = solver.frcg_minimize(_oracle_fdf(ref), np.zeros(w.P), **solver.BACKEND)
    x, rep = be.setupProblemAndOptimize()
    assert rep["final_cost"] < rep["initial_cost"]
    assert abs(rep["final_cost"] - rep_ref["final_cost"]) < 2e-3 * abs(rep_ref["final_cost"])
    assert rep["initial_cost"] == pytest.approx(rep_ref["initial_cost"], rel=1e-6)
    assert abs(rep["iterations"] - rep_ref["iterations"]) <= 3


def test_solve_reports_evaluator_errors(hip):
    p = synth.frontend_packet(2000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=1)
    fe = hip.reference_shaped.FrontendEvaluator(p.W, p.H, p.lut)
    with pytest.raises(hip.CmaxHipError):  # no packet: the functor fails, the solve returns the status
        fe.setupProblemAndOptimize(np.zeros(3))
