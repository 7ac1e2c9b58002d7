"""-m gpu: seeded random configurations of both paths against the CPU oracle -- sensor / panorama sizes that are not
multiples of the tile sizes, every blur radius the kernels accept, batch sizes and sampling rates that leave ragged
batches, both measures, both spline orders, every fixed-knot count, empty / partial global maps, and sequences of
evaluations that alternate cost-only and gradient calls (ping-pong buffers, image reuse, tile occupancy)."""
import os

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, backend_fuzz_config, backend_fuzz_points, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu
N_FE = int(os.environ.get("CMX_FUZZ_FE", "10"))   # CMX_FUZZ_FE=300 CMX_FUZZ_BE=300 for a deeper one-off sweep
N_BE = int(os.environ.get("CMX_FUZZ_BE", "12"))


@pytest.mark.parametrize("seed", range(N_FE))
def test_frontend_random_configuration(hip, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.integers(48, 400))
    H = int(rng.integers(40, 300))
    f = float(rng.uniform(0.6, 1.4) * max(W, H))
    N = int(rng.integers(1, 30_000))
    batch = int(rng.choice([1, 7, 64, 100, 257]))
    sigma = float(rng.choice([0.0, 0.5, 1.0, 1.7, 3.0]))
    measure = int(rng.choice([_lib.VARIANCE, _lib.MEAN_SQUARE]))
    p = synth.frontend_packet(N, W, H, f, f, (W - 1) / 2, (H - 1) / 2, T=float(rng.uniform(0.01, 0.08)), seed=seed)
    fe = hip.reference_shaped.FrontendEvaluator(W, H, p.lut)
    if seed % 3 != 2:
        fe.set_fast_path()
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, batch, sigma, measure)
    ref = oracle.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, batch, sigma, measure)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    x = np.zeros(3)
    for step in range(5):
        want = bool(rng.integers(0, 2)) or step == 0
        if rng.random() < 0.7:
            x = p.omega_true * rng.uniform(0, 1.5) + rng.normal(0, 0.3, 3)   # else: same point again (image reuse)
        c_ref, g_ref = ref.eval(x, want)
        c, g = fe.eval(x, want)
        assert rel_scalar(c, c_ref) < RTOL, (seed, step, W, H, N, batch, sigma, measure)
        if want:
            assert rel_vec(g, g_ref) < RTOL, (seed, step, W, H, N, batch, sigma, measure)


@pytest.mark.parametrize("seed", range(N_BE))
def test_backend_random_configuration(hip, oracle, seed):
    rng, k, w, IG = backend_fuzz_config(seed)
    be = hip.reference_shaped.BackendEvaluator(k["W"], k["H"], w.lut, k["Wp"], k["Hp"])
    fast = seed % 4 != 3
    if fast:
        be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, k["order"], w.knots_init, w.start_ns, w.dt_ns, k["nf"], w.t_next_win_beg_ns, k["batch"],
                  k["rate"], k["sigma"], k["measure"], IG)
    ref = oracle.Backend(k["W"], k["H"], w.lut, k["Wp"], k["Hp"], k["order"], k["batch"], k["rate"], k["sigma"], k["measure"])
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, k["nf"], w.t_next_win_beg_ns, IG)
    for step, (want, x) in enumerate(backend_fuzz_points(rng, k["P"])):
        c_ref, g_ref = ref.eval(x, want)
        c, g = be.eval(x, want)
        tag = (seed, step, fast, sorted(k.items()))
        assert rel_scalar(c, c_ref) < RTOL, tag
        if want:
            # north_star's tolerance, nothing on top: 1e-5 of the gradient's max-norm
            assert rel_vec(g, g_ref) < RTOL, tag + (np.abs(g - g_ref).max(), np.abs(g_ref).max())
    if IG is not None:
        assert rel_scalar(be.alpha, ref.alpha) < RTOL
