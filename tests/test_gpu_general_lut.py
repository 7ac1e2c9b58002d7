"""-m gpu: a bearing table whose z column is NOT all ones (e.g. unit-norm rays).  The fast kernels then cannot use the
16-byte (x, y) table nor the per-event bearing streams and take the general three-double path; the reference indexes
whatever `precomputed_bearing_vectors` holds (local_image_warped_events.cpp:100, event_pano_warper.cpp:265)."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


def _scaled(lut, seed):
    l = np.array(lut, np.float64).reshape(-1, 3).copy()
    l *= np.random.default_rng(seed).uniform(0.5, 2.0, (len(l), 1))   # any positive scale per pixel; z != 1 now
    return l.reshape(-1)


@pytest.mark.parametrize("fast", [True, False])
def test_frontend_general_table(hip, oracle, fast):
    p = synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=41)
    lut = _scaled(p.lut, 1)
    fe = hip.FrontendEvaluator(p.W, p.H, lut)
    fe.set_fast_path() if fast else fe.set_reference_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    ref = oracle.Frontend(p.W, p.H, lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    for x in (np.array([0.3, -0.5, 0.2]), np.array([0.7, -1.0, 0.5])):
        c_ref, g_ref = ref.eval(x)
        c, g = fe.eval(x)
        assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
        assert rel_scalar(fe.eval(x, False)[0], c_ref) < RTOL


@pytest.mark.parametrize("fast", [True, False])
def test_backend_general_table(hip, oracle, fast):
    w = synth.backend_window(50_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 8, 3, 0.25, seed=42)
    lut = _scaled(w.lut, 2)
    be = hip.BackendEvaluator(w.W, w.H, lut, w.Wp, w.Hp)
    be.set_fast_path() if fast else be.set_reference_path()
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                  w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
    ref = oracle.Backend(w.W, w.H, lut, w.Wp, w.Hp, w.order, w.batch, w.sample_rate, w.sigma, oracle.VARIANCE)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    d = 0.01 * np.sin(np.arange(w.P) + 0.5)
    c_ref, g_ref = ref.eval(d)
    c, g = be.eval(d)
    assert rel_scalar(c, c_ref) < RTOL and rel_vec(g, g_ref) < RTOL
    assert rel_scalar(be.eval(d, False)[0], c_ref) < RTOL
