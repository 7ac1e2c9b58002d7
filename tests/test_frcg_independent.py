"""CPU: the product's FR-CG driver (cmx_frcg_minimize: host C++, cmx_frcg_sm.hpp + cmx_solver.cpp) against a second,
independently written restatement of GSL's conjugate_fr + the reference's stopping rules (oracle/frcg.py, Python),
CALL FOR CALL: both are run over the same deterministic functor and must ask for the same evaluations (cost-only or
with gradient) at bitwise the same points, and return the same iterate, counts and costs.

Neither side is GSL (un-vendored, absent): this does not pin the reference's optimiser, it removes "the driver is
only ever compared with itself" (src/frontend/local_optim_contrast_gsl.cpp:74-233,
src/backend/global_optim_contrast_gsl.cpp:15-145 are the loops both restate)."""
import numpy as np
import pytest

from cmax_slam_amd import solver, synth
from oracle import frcg


def _traced(fdf):
    calls = []

    def wrapped(x, want_grad):
        x = np.asarray(x, np.float64)
        calls.append((bool(want_grad), x.tobytes()))
        return fdf(x, want_grad)
    return wrapped, calls


def _both(fdf, x0, **kw):
    f1, calls_cpp = _traced(fdf)
    x_cpp, rep_cpp = solver.frcg_minimize(f1, np.array(x0, np.float64), **kw)
    f2, calls_py = _traced(fdf)
    x_py, rep_py = frcg.minimize(f2, list(map(float, x0)), **kw)
    return (x_cpp, rep_cpp, calls_cpp), (np.array(x_py), rep_py, calls_py)


def _assert_identical(a, b):
    (xa, ra, ca), (xb, rb, cb) = a, b
    assert len(ca) == len(cb), (len(ca), len(cb), ra, rb)
    for k, (u, v) in enumerate(zip(ca, cb)):
        assert u[0] == v[0], "call %d: cost-only vs gradient differs" % k
        assert u[1] == v[1], "call %d: evaluation point differs: %r vs %r" % (k, np.frombuffer(u[1]), np.frombuffer(v[1]))
    assert xa.tobytes() == xb.tobytes()
    for key in ("iterations", "status", "n_f", "n_df"):
        assert ra[key] == rb[key], (key, ra, rb)
    assert ra["initial_cost"] == rb["initial_cost"] and ra["final_cost"] == rb["final_cost"]


def test_quadratic_same_call_sequence():
    A = np.diag([1.0, 4.0, 9.0, 0.5])
    b = np.array([1.0, -2.0, 3.0, 0.25])
    fdf = lambda x, wg: (float(0.5 * x @ A @ x - b @ x), (A @ x - b) if wg else None)
    a, c = _both(fdf, np.zeros(4), step_size=0.1, tol=0.05, epsabs_grad=1e-8, tolfun=1e-13, max_iterations=100)
    _assert_identical(a, c)
    assert a[1]["iterations"] > 3


def test_rosenbrock_same_call_sequence():
    def fdf(x, wg):
        f = (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2
        g = np.array([-2 * (1 - x[0]) - 400 * x[0] * (x[1] - x[0] ** 2), 200 * (x[1] - x[0] ** 2)])
        return float(f), (g if wg else None)
    a, c = _both(fdf, [-1.2, 1.0], step_size=0.01, tol=1e-4, epsabs_grad=1e-6, tolfun=0.0, max_iterations=300)
    _assert_identical(a, c)
    assert a[1]["iterations"] == 300 or a[1]["final_cost"] < 1e-3


def test_stationary_start_and_flat_direction():
    fdf = lambda x, wg: (float(x @ x), 2 * x if wg else None)
    a, c = _both(fdf, np.zeros(2))
    _assert_identical(a, c)
    assert a[1]["status"] == frcg.GSL_ENOPROG
    # a direction along which the trial point cannot move (x + dx == x): intermediate_point's fast exit
    fdf2 = lambda x, wg: (float(x[0]), np.array([1.0, 0.0]) if wg else None)
    a, c = _both(fdf2, [1e17, 0.0], max_iterations=3)
    assert a[1]["status"] == frcg.GSL_ENOPROG and a[1]["n_df"] == 2
    _assert_identical(a, c)


def test_frontend_functor_same_call_sequence(oracle):
    p = synth.frontend_packet(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=33)
    fe = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)

    def fdf(x, wg):
        c, g = fe.eval(x, wg)
        return -c, (-g if wg else None)
    a, c = _both(fdf, np.zeros(3), **solver.FRONTEND)
    _assert_identical(a, c)
    assert frcg.FRONTEND == solver.FRONTEND
    assert 2 <= a[1]["iterations"] <= 50 and a[1]["final_cost"] < a[1]["initial_cost"]


@pytest.mark.parametrize("order,K,nf,T", [(2, 5, 1, 0.2), (4, 10, 3, 0.35)])
def test_backend_functor_same_call_sequence(oracle, order, K, nf, T):
    w = synth.backend_window(20_003, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, order, K, nf, T, seed=5)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)

    def fdf(x, wg):
        c, g = ref.eval(x, wg)
        return -c, (-g if wg else None)
    a, c = _both(fdf, np.zeros(w.P), **solver.BACKEND)
    _assert_identical(a, c)
    assert frcg.BACKEND == solver.BACKEND
    assert a[1]["final_cost"] < a[1]["initial_cost"]
