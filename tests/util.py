"""Shared helpers for the parity tests."""
import numpy as np

# north_star tolerance: results match the reference CPU IWE / variance / gradient within 1e-5 relative (fp32
# accumulators; the only difference between the HIP path and the oracle is the order of the fp32 atomic adds
# and libm-vs-ocml ulps in atan2/asin/sin/cos).
RTOL = 1e-5


def rel_img(a, b):
    """max |a-b| relative to the image's own scale."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel_vec(a, b):
    """max-norm error relative to the max-norm of the reference vector."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel_scalar(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def grad_cancellation_scale(iwe, planes, measure):
    """max_k (2/N) sum_px |I - mu| |D_k - mean(D_k)|: the sum of the magnitudes of the terms the reference's gradient
    formula adds up (local_focus_funcs.cpp:36-40, global_focus_funcs.cpp:39-43).  Near a stationary point the gradient
    is a small difference of these terms, and the fp32 rounding of the images -- in the reference as much as here --
    is relative to THIS scale, not to the gradient itself."""
    I = np.asarray(iwe, np.float64)
    variance = measure == 0
    dev = np.abs(I - (I.mean() if variance else 0.0))
    best = 0.0
    for D in planes:
        D = np.asarray(D, np.float64)
        best = max(best, 2.0 * float(np.sum(dev * np.abs(D - (D.mean() if variance else 0.0)))) / I.size)
    return best


def backend_fuzz_config(seed):
    """The seeded random back-end configuration of tests/test_gpu_fuzz.py (also replayed by tests/exact_noise.py and
    tests/oracle_order_noise.py).  Returns (rng, cfg dict, window, IG): the rng continues with the evaluation sequence."""
    from cmax_slam_amd import synth
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
    measure = int(rng.choice([0, 1]))
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
    cfg = dict(W=W, H=H, Wp=Wp, Hp=Hp, order=order, K=K, nf=nf, N=N, batch=batch, rate=rate, sigma=sigma, measure=measure,
               kind=kind, P=3 * (K - nf))
    return rng, cfg, w, IG


def backend_fuzz_points(rng, P, steps=5):
    """The evaluation sequence of a fuzz configuration: (want_grad, x) per step (same draws as the original loop)."""
    x = np.zeros(P)
    out = []
    for step in range(steps):
        want = bool(rng.integers(0, 2)) or step == 0
        if rng.random() < 0.7:
            x = rng.normal(0, float(rng.choice([0.002, 0.02, 0.1])), P)
        out.append((want, x.copy()))
    return out
