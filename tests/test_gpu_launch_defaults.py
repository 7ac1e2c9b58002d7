"""-m gpu: the reference's OWN back-end operating points at full size against the oracle.

launch/ijrr.launch:27-33 and launch/ecrot_handheld.launch:28-34: linear spline (spline_degree 1 -> So3Spline<2>), dt_knots
0.05 s, sliding window 0.2 s -> 5 control poses, P = 15 (no fixed pose: the first window) or 12 (one fixed pose), event
batches of 100, blur sigma 1, variance; panorama 1024x512 (DAVIS 240x180) or 4096x2048 (1280x720 sensor).  bench.py's
`backend.launch_defaults` times exactly these shapes; here each is compared with the CPU oracle on all of its events:
contrast, gradient, both vote planes, at zero increments and at a solve-sized step, and the solve's outcome."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, solver, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu

SHAPES = {"ijrr": (240, 180, 200.0, 1024, 512, 0.1), "ecrot_handheld": (1280, 720, 1000.0, 4096, 2048, 0.2)}


def _window(launch, n_ev, nf):
    W, H, f, Wp, Hp, stride = SHAPES[launch]
    return synth.backend_window(n_ev, W, H, f, f, (W - 1) / 2.0, (H - 1) / 2.0, Wp, Hp, 2, 5, nf, 0.2, dt_knots=0.05,
                                seed=synth.SEED0 + 40 + nf, win_stride=stride)


@pytest.mark.parametrize("launch,n_ev,nf", [("ijrr", 200_000, 1), ("ijrr", 1_000_000, 0), ("ecrot_handheld", 200_000, 0),
                                            ("ecrot_handheld", 1_000_000, 1)])
def test_launch_default_shape_vs_oracle(hip, oracle, launch, n_ev, nf):
    w = _window(launch, n_ev, nf)
    assert (w.order, w.K, w.P, w.batch, w.sigma) == (2, 5, 15 - 3 * nf, 100, 1.0) and w.dt_ns == 50_000_000
    assert w.t_ns[-1] - w.t_ns[0] <= 200_000_000
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    rng = np.random.default_rng(31)
    for d in (np.zeros(w.P), rng.normal(0, 0.003, w.P)):
        c_ref, g_ref = ref.eval(d)
        c, g = be.eval(d)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL, (launch, n_ev, nf, c, c_ref)
        assert rel_scalar(be.eval(d, False)[0], c_ref) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), ref.IL_old) < RTOL and rel_img(be.get_plane(_lib.PLANE_IL_NEW), ref.IL_new) < RTOL
    # both vote planes are in use (stride < window for ijrr; the whole window is "old" for ecrot_handheld's stride = window)
    assert ref.IL_old.sum() > 0 and (ref.IL_new.sum() > 0) == (launch == "ijrr")
    if n_ev == 200_000 and launch == "ijrr":   # (the oracle blurs 14-17 planes per evaluation: seconds at 4096x2048) the window's solve, host C++ driver over the HIP evaluator vs the same driver over the oracle
        x, rep = be.setupProblemAndOptimize()

        def fdf(xx, wg):
            cc, gg = ref.eval(xx, wg)
            return -cc, (-gg if wg else None)
        x_ref, rep_ref = solver.frcg_minimize(fdf, np.zeros(w.P), **solver.BACKEND)
        assert rep["initial_cost"] == pytest.approx(rep_ref["initial_cost"], rel=1e-6)
        assert abs(rep["final_cost"] - rep_ref["final_cost"]) < 5e-3 * abs(rep_ref["final_cost"]), (rep, rep_ref)
        assert rep["final_cost"] < rep["initial_cost"]
