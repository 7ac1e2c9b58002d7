"""-m gpu: how an evaluation hands its results to the host, in every combination the options allow.

  CMX_OPT_SPIN_WAIT = 1 (default): the host spins on a completion ticket + checksum the finalize step writes to mapped host
      memory; 0: plain hipStreamSynchronize.
  CMX_OPT_TAIL_FINALIZE = 1: that finalize step runs in the last-arriving workgroup of the evaluation's last kernel
      (write-through partial sums, XCD-sharded tickets); 0: its own one-workgroup launch.
All four combinations must return what the oracle returns, for both ends, for cost-only and gradient evaluations, across a
sequence that alternates them (ticket numbering, ping-pong buffers, image reuse, speculative adjoint pass)."""
import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spin", [0, 1])
@pytest.mark.parametrize("tail", [0, 1])
def test_frontend_every_hand_off(hip, oracle, spin, tail):
    p = synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=71)
    fe = hip.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_fast_path()
    fe.set_option(_lib.OPT_SPIN_WAIT, spin)
    fe.set_option(_lib.OPT_TAIL_FINALIZE, tail)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, oracle.VARIANCE)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    rng = np.random.default_rng(1)
    om = np.zeros(3)
    for step in range(12):
        want = step % 3 != 1
        if step % 4 != 3:                      # every fourth step repeats the point: df after f reuses the image
            om = p.omega_true * rng.uniform(0, 1.2) + rng.normal(0, 0.1, 3)
        c_ref, g_ref = ref.eval(om, want)
        c, g = fe.eval(om, want)
        assert rel_scalar(c, c_ref) < RTOL, (spin, tail, step)
        if want:
            assert rel_vec(g, g_ref) < RTOL, (spin, tail, step)
    x, rep = fe.setupProblemAndOptimize(np.zeros(3))
    assert rep["final_cost"] < rep["initial_cost"]
    st = fe.stats()
    assert st["reuse_hits"] > 0 and st["spec_hits"] > 0   # the solver's f-then-df pairs found image and Jt ready


@pytest.mark.parametrize("spin", [0, 1])
@pytest.mark.parametrize("tail", [0, 1])
def test_backend_every_hand_off(hip, oracle, spin, tail):
    w = synth.backend_window(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 4, 10, 3, 0.35, seed=72)
    be = hip.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_fast_path()
    be.set_option(_lib.OPT_SPIN_WAIT, spin)
    be.set_option(_lib.OPT_TAIL_FINALIZE, tail)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    rng = np.random.default_rng(2)
    d = np.zeros(w.P)
    for step in range(10):
        want = step % 3 != 1
        if step % 4 != 3:
            d = rng.normal(0, 0.01, w.P)
        c_ref, g_ref = ref.eval(d, want)
        c, g = be.eval(d, want)
        assert rel_scalar(c, c_ref) < RTOL, (spin, tail, step)
        if want:
            assert rel_vec(g, g_ref) < RTOL, (spin, tail, step)
    x, rep = be.setupProblemAndOptimize()
    assert rep["final_cost"] < rep["initial_cost"]
