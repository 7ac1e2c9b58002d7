"""oracle/gen_cv_fixtures.py -- blur / contrast fixtures from a SECOND implementation of the OpenCV rules.

TEST INFRASTRUCTURE.  oracle/cv_ops.c restates cv::GaussianBlur(Size(0,0), sigma) / meanStdDev / mean / MatExpr as the
reference uses them (src/frontend/local_image_warped_events.cpp:34-37, src/frontend/local_focus_funcs.cpp:9-44,
src/backend/global_focus_funcs.cpp:11-47) from OpenCV's documented behaviour; OpenCV itself is absent here, so that
restatement is "parity unpinned".  This script computes the same quantities with scipy.ndimage + numpy in float64:

    blur      scipy.ndimage.gaussian_filter1d(axis 1 then axis 0, mode='mirror' (= BORDER_REFLECT_101),
              radius = (cvRound(8 sigma + 1) | 1) // 2)       -- the ksize rule is the one thing taken from OpenCV
    variance  np.mean((B - B.mean())**2)                      contrast_Variance:      sigma^2, population
    gradient  np.mean(2 (B - mu) (D_k - D_k.mean()))          local_focus_funcs.cpp:36-40
    mean sq.  np.mean(B**2), 2 np.mean(B D_k)                 contrast_MeanSquare

and writes tests/golden/cv_scipy.npz (inputs + float64 results).  tests/test_oracle_cv_scipy.py checks oracle/cv_ops.c
against the committed file AND against scipy run live.  It does not pin the reference (scipy is not OpenCV): it gives
the restatement a second, independently written implementation that agrees with it to fp32 rounding.

    python oracle/gen_cv_fixtures.py        (re)writes tests/golden/cv_scipy.npz
"""
import os

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "cv_scipy.npz")
SIGMAS = (0.5, 0.8, 1.0, 1.1, 1.7, 2.0, 3.0)   # 1.1: scipy's own truncate rule would pick radius 4, OpenCV's ksize rule 5


def cv_radius(sigma):
    """GaussianBlur(Size(0,0)) on CV_32F: ksize = cvRound(sigma*4*2 + 1) | 1 (cvRound = round half to even)."""
    return (int(np.rint(sigma * 8 + 1)) | 1) // 2


def blur64(img, sigma):
    r = cv_radius(sigma)
    a = np.asarray(img, np.float64)
    a = ndimage.gaussian_filter1d(a, sigma, axis=1, mode="mirror", radius=r)
    return ndimage.gaussian_filter1d(a, sigma, axis=0, mode="mirror", radius=r)


def kernel64(sigma):
    r = cv_radius(sigma)
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * x * x / (sigma * sigma))
    return k / k.sum()


def contrast64(B, D, measure):
    """B: blurred image, D: list of blurred derivative planes; returns (contrast, gradient)."""
    B = np.asarray(B, np.float64)
    if measure == 0:
        mu = B.mean()
        return float(np.mean((B - mu) ** 2)), np.array([np.mean(2.0 * (B - mu) * (d - d.mean())) for d in D])
    return float(np.mean(B * B)), np.array([2.0 * np.mean(B * d) for d in D])


def make_inputs():
    rng = np.random.default_rng(20240314)
    shapes = [(37, 53), (48, 64), (9, 120), (5, 7)]
    imgs, planes = [], []
    for H, W in shapes:
        img = np.zeros((H, W), np.float32)
        n = max(8, H * W // 6)
        ys, xs = rng.integers(0, H, n), rng.integers(0, W, n)
        np.add.at(img, (ys, xs), rng.uniform(0, 3, n).astype(np.float32))   # sparse, event-image-like
        img[0, :] += 1.0   # mass on the border rows / columns: exercises the reflection
        img[:, -1] += 0.5
        imgs.append(img)
        planes.append(rng.normal(0, 1, (3, H, W)).astype(np.float32) * (img > 0))
    return imgs, planes


def main():
    imgs, planes = make_inputs()
    out = {"sigmas": np.array(SIGMAS), "n_images": np.array(len(imgs))}
    for i, (img, pl) in enumerate(zip(imgs, planes)):
        out["img%d" % i] = img
        out["planes%d" % i] = pl
        for j, s in enumerate(SIGMAS):
            if 2 * cv_radius(s) + 1 > 2 * min(img.shape) - 1:   # the kernel would reflect more than once: not the reference's regime
                continue
            B = blur64(img, s)
            D = [blur64(p, s) for p in pl]
            out["blur%d_%d" % (i, j)] = B
            for m in (0, 1):
                c, g = contrast64(B, D, m)
                out["c%d_%d_%d" % (i, j, m)] = np.array(c)
                out["g%d_%d_%d" % (i, j, m)] = g
    for j, s in enumerate(SIGMAS):
        out["kernel_%d" % j] = kernel64(s)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
