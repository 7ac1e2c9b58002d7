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
