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
