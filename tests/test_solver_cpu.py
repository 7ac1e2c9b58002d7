"""CPU: the restated FR-CG driver (cmx_frcg_minimize, host C++ in libcmaxhip.so) on known functions and on the
CPU oracle's cost functor.  (GSL itself is absent: the restatement is unpinned against GSL, see cmx_frcg_sm.hpp.)"""
import numpy as np
import pytest

from cmax_slam_amd import solver, synth


def test_quadratic_converges():
    A = np.diag([1.0, 4.0, 9.0])
    b = np.array([1.0, -2.0, 3.0])
    fdf = lambda x, wg: (0.5 * x @ A @ x - b @ x, (A @ x - b) if wg else None)
    x, rep = solver.frcg_minimize(fdf, np.zeros(3), tolfun=1e-12, epsabs_grad=1e-8, max_iterations=200)
    np.testing.assert_allclose(x, np.linalg.solve(A, b), atol=1e-5)
    assert rep["n_f"] > 0 and rep["n_df"] > 0 and rep["final_cost"] < rep["initial_cost"]


def test_rosenbrock_descends():
    def fdf(x, wg):
        f = (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2
        g = np.array([-2 * (1 - x[0]) - 400 * x[0] * (x[1] - x[0] ** 2), 200 * (x[1] - x[0] ** 2)])
        return f, (g if wg else None)
    x, rep = solver.frcg_minimize(fdf, np.array([-1.2, 1.0]), step_size=0.01, tol=1e-4, tolfun=0.0, epsabs_grad=1e-6,
                                  max_iterations=2000)
    assert rep["final_cost"] < 1e-3 and np.abs(x - 1).max() < 0.05


def test_zero_gradient_start_reports_no_progress():
    fdf = lambda x, wg: (float(x @ x), 2 * x if wg else None)
    x, rep = solver.frcg_minimize(fdf, np.zeros(2))
    assert rep["status"] == 27 and rep["iterations"] == 1  # GSL_ENOPROG: pnorm == 0


def test_frontend_solve_on_the_oracle_recovers_the_motion(oracle):
    p = synth.frontend_packet(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=33)
    fe = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)

    def fdf(x, wg):
        c, g = fe.eval(x, wg)
        return -c, (-g if wg else None)
    x, rep = solver.frcg_minimize(fdf, np.zeros(3), **solver.FRONTEND)
    assert rep["final_cost"] < rep["initial_cost"]
    # the reference's stagnation rule (|1 - c_new/c_old| < 1e-4) stops before the weakly observable roll rate
    # has converged; pan/tilt rates and the contrast itself are recovered
    assert np.abs(x[:2] - p.omega_true[:2]).max() < 0.05, (x, rep)
    assert -rep["final_cost"] > 0.95 * fe.eval(p.omega_true, False)[0]
    assert 2 <= rep["iterations"] <= 50
    assert rep["n_df"] <= rep["n_f"] + 1  # GSL's pattern: f-only trial points, df at accepted points
