"""CPU: oracle/cv_ops.c (the restated cv::GaussianBlur / meanStdDev / mean / MatExpr rules; OpenCV is absent, so that
restatement is parity-unpinned) against a second, independently written implementation -- scipy.ndimage + numpy in
float64 (oracle/gen_cv_fixtures.py) -- both through the committed fixture tests/golden/cv_scipy.npz and live.
Not a pin of the reference (scipy is not OpenCV): two agreeing restatements instead of one."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import gen_cv_fixtures as gen  # noqa: E402

FIX = np.load(os.path.join(HERE, "golden", "cv_scipy.npz"))
N_IMG = int(FIX["n_images"])
SIGMAS = [float(s) for s in FIX["sigmas"]]


def test_fixture_matches_scipy_run_live():
    imgs, planes = gen.make_inputs()
    assert tuple(SIGMAS) == gen.SIGMAS and len(imgs) == N_IMG
    for i in range(N_IMG):
        assert np.array_equal(FIX["img%d" % i], imgs[i]) and np.array_equal(FIX["planes%d" % i], planes[i])
        for j, s in enumerate(SIGMAS):
            if "blur%d_%d" % (i, j) in FIX:
                np.testing.assert_allclose(gen.blur64(imgs[i], s), FIX["blur%d_%d" % (i, j)], rtol=0, atol=1e-12)


@pytest.mark.parametrize("j", range(len(SIGMAS)))
def test_kernel_size_and_taps(oracle, j):
    s = SIGMAS[j]
    k = oracle.gauss_kernel(s)
    assert len(k) == 2 * gen.cv_radius(s) + 1 == len(FIX["kernel_%d" % j])
    np.testing.assert_allclose(k, FIX["kernel_%d" % j], rtol=0, atol=6e-8)   # fp32 cast of the fp64 kernel
    assert abs(float(np.sum(k.astype(np.float64))) - 1.0) < 3e-7


@pytest.mark.parametrize("i", range(N_IMG))
def test_gaussian_blur_matches_scipy_mirror(oracle, i):
    img = FIX["img%d" % i]
    done = 0
    for j, s in enumerate(SIGMAS):
        key = "blur%d_%d" % (i, j)
        if key not in FIX:
            continue
        got = oracle.gaussian_blur(img, s).astype(np.float64)
        ref = FIX[key]
        assert np.abs(got - ref).max() <= 4e-7 * max(1.0, np.abs(ref).max()), (i, s)   # fp32 taps + fp32 accumulation
        done += 1
    assert done >= 1


@pytest.mark.parametrize("measure", [0, 1])
@pytest.mark.parametrize("i", range(N_IMG))
def test_contrast_and_gradient_match_numpy(oracle, i, measure):
    img, planes = FIX["img%d" % i], FIX["planes%d" % i]
    for j, s in enumerate(SIGMAS):
        if "blur%d_%d" % (i, j) not in FIX:
            continue
        B = oracle.gaussian_blur(img, s)
        D = np.stack([oracle.gaussian_blur(p, s) for p in planes])
        c, g = oracle.contrast(B, D, measure)
        c_ref, g_ref = float(FIX["c%d_%d_%d" % (i, j, measure)]), FIX["g%d_%d_%d" % (i, j, measure)]
        assert abs(c - c_ref) <= 2e-6 * abs(c_ref), (i, s, c, c_ref)
        # the gradient is a sum of products of two fp32-rounded images: compare on the scale of the terms it adds up
        dev = np.abs(B.astype(np.float64) - (B.mean() if measure == 0 else 0.0))
        scale = max(2.0 * float(np.mean(dev * np.abs(d.astype(np.float64) - (d.mean() if measure == 0 else 0.0)))) for d in D)
        assert np.abs(g - g_ref).max() <= 2e-6 * scale, (i, s, g, g_ref)
