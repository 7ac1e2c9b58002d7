"""-m gpu: seeded random configurations of both paths against the CPU oracle -- sensor / panorama sizes that are not
multiples of the tile sizes, every blur radius the kernels accept, batch sizes and sampling rates that leave ragged
batches, both measures, both spline orders, every fixed-knot count, empty / partial global maps, and sequences of
evaluations that alternate cost-only and gradient calls (ping-pong buffers, image reuse, tile occupancy)."""
import os

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, grad_cancellation_scale, rel_scalar, rel_vec

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
    fe = hip.FrontendEvaluator(W, H, p.lut)
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
    rng = np.random.default_rng(2000 + seed)
    W, H = int(rng.integers(64, 260)), int(rng.integers(48, 200))
    f = float(rng.uniform(0.7, 1.3) * max(W, H))
    Hp = int(rng.choice([96, 200, 256, 300, 512]))
    Wp = 2 * Hp if seed % 4 else int(rng.choice([130, 640, 1000]))
    order = int(rng.choice([2, 4]))
    K = order + int(rng.integers(0, 6))
    nf = int(rng.integers(0, K))            # 0 .. K-1 fixed knots (at least one free)
    dt_knots = float(rng.choice([0.02, 0.05]))
    T = float(rng.uniform(0.3, 1.0)) * (K - order + 1) * dt_knots
    N = int(rng.integers(200, 40_000))
    batch = int(rng.choice([3, 50, 100, 128]))
    rate = int(rng.choice([1, 1, 2, 5]))
    sigma = float(rng.choice([0.0, 0.8, 1.0, 2.0, 3.0]))
    measure = int(rng.choice([_lib.VARIANCE, _lib.MEAN_SQUARE]))
    w = synth.backend_window(N, W, H, f, f, (W - 1) / 2, (H - 1) / 2, Wp, Hp, order, K, nf, T, dt_knots=dt_knots,
                             seed=300 + seed, knot_sigma=float(rng.choice([0.01, 0.05, 0.15])))
    IG = None
    kind = seed % 3
    if kind:
        IG = np.zeros((Hp, Wp), np.float32)
        for _ in range(3 if kind == 1 else 12):   # a few blobs (partial map) or many (most of the band covered)
            cx, cy = rng.integers(0, Wp), rng.integers(0, Hp)
            yy, xx = np.mgrid[0:Hp, 0:Wp]
            IG += (rng.uniform(0.5, 4) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / rng.uniform(20, 400))).astype(np.float32)
        IG[IG < 0.05] = 0
    be = hip.BackendEvaluator(W, H, w.lut, Wp, Hp)
    fast = seed % 4 != 3
    if fast:
        be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, order, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, batch, rate, sigma,
                  measure, IG)
    ref = oracle.Backend(W, H, w.lut, Wp, Hp, order, batch, rate, sigma, measure)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
    P = 3 * (K - nf)
    x = np.zeros(P)
    for step in range(5):
        want = bool(rng.integers(0, 2)) or step == 0
        if rng.random() < 0.7:
            x = rng.normal(0, float(rng.choice([0.002, 0.02, 0.1])), P)
        c_ref, g_ref = ref.eval(x, want)
        c, g = be.eval(x, want)
        tag = (seed, step, W, H, Wp, Hp, order, K, nf, N, batch, rate, sigma, measure, kind, fast)
        assert rel_scalar(c, c_ref) < RTOL, tag
        if want:
            # 1e-5 of the gradient, plus the fp32 noise floor of the formula itself where the gradient is a small
            # difference of large sums (4 of 250 random configurations, all with sigma >= 2 near a stationary point,
            # deviate by 1.4e-5..2.3e-5 of |g| in BOTH the adjoint and the reference-shaped GPU path)
            iwe_ref, planes_ref = ref.iwe(x, planes=True)
            floor = 3e-7 * grad_cancellation_scale(iwe_ref, planes_ref, measure)
            err = np.abs(g - g_ref).max()
            assert err <= RTOL * np.abs(g_ref).max() + floor, tag + (err, np.abs(g_ref).max(), floor)
    if IG is not None:
        assert rel_scalar(be.alpha, ref.alpha) < RTOL
