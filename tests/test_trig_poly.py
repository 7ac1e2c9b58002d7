"""CPU: the polynomial forms of cmax_slam_amd/csrc/cmx_trig.hpp (the back end's fp64 atan2 / asin), re-evaluated in numpy with the
coefficients parsed from the header, against 50-digit values (mpmath).  The device code uses fused multiply-adds and hardware
reciprocal / rsqrt seeds with Newton steps where numpy divides and takes square roots; what is pinned here is the coefficients, the
range reductions and the octant / sign logic.  (End to end: the -m gpu parity suite and tests/exact_noise.py.)"""
import os
import re

import mpmath as mp
import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cmax_slam_amd", "csrc", "cmx_trig.hpp")


def _coeffs(name):
    src = open(HDR).read()
    m = re.search(r"double %s\[(\d+)\] = \{(.*?)\};" % name, src, re.S)
    vals = [float(v) for v in m.group(2).replace("\n", " ").split(",")]
    assert len(vals) == int(m.group(1))
    return vals


def _horner(c, r):
    p = np.full_like(r, c[-1])
    for k in range(len(c) - 2, -1, -1):
        p = p * r + c[k]
    return p


def lean_atan2(y, x):
    c = _coeffs("kAtanQ")
    ax, ay = np.abs(x), np.abs(y)
    hi, lo = np.maximum(ax, ay), np.minimum(ax, ay)
    a = np.where(hi > 0, lo / np.where(hi > 0, hi, 1.0), 0.0)
    r = a * a
    t = a + a * r * _horner(c, r)
    t = np.where(ay > ax, np.pi / 2 - t, t)
    t = np.where(x < 0, np.pi - t, t)
    return np.copysign(t, y)


def lean_asin(t):
    c = _coeffs("kAsinQ")
    at = np.minimum(np.abs(t), 1.0)
    big = at > 0.5
    rb = at * -0.5 + 0.5
    r = np.where(big, rb, at * at)
    s = np.where(big, np.sqrt(rb), at)
    u = s + s * r * _horner(c, r)
    return np.copysign(np.where(big, np.pi / 2 - 2 * u, u), t)


def _ulps(approx, exact_mp):
    mp.mp.dps = 50
    ex = np.array([float(v) for v in exact_mp])
    err = np.array([float(abs(mp.mpf(float(a)) - e)) for a, e in zip(approx, exact_mp)])
    return err / np.spacing(np.maximum(np.abs(ex), 1e-300))


def test_atan2_all_octants_within_a_few_ulp():
    mp.mp.dps = 50
    rng = np.random.default_rng(1)
    n = 6000
    y = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n)
    x = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n)
    y[:8] = [0.0, 1.0, -1.0, 1.0, -1.0, 0.0, 2.5, -2.5]      # axes and diagonals
    x[:8] = [1.0, 0.0, 0.0, 1.0, -1.0, -1.0, 2.5, -2.5]
    got = lean_atan2(y, x)
    ref = [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y, x)]
    u = _ulps(got, ref)
    assert u.max() < 4.0, (u.max(), y[np.argmax(u)], x[np.argmax(u)])
    assert lean_atan2(np.array([0.0]), np.array([0.0]))[0] == 0.0
    tiny = lean_atan2(np.array([3e-310, 0.0, 1e-320]), np.array([4e-310, 5e-315, 0.0]))      # subnormal rays (the pole)
    assert np.allclose(tiny, np.arctan2([3e-310, 0.0, 1e-320], [4e-310, 5e-315, 0.0]), rtol=0, atol=1e-15)
    assert np.abs(got - np.arctan2(y, x)).max() < 1e-15


def test_asin_both_branches_within_a_few_ulp():
    mp.mp.dps = 50
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(-1, 1, 6000), [0.0, 0.5, -0.5, 0.5000000001, 1.0, -1.0, 1.0 - 1e-12, 1e-9, -1e-300],
                        np.sign(rng.normal(size=500)) * (1 - 10.0 ** rng.uniform(-15, -1, 500))])
    got = lean_asin(t)
    ref = [mp.asin(mp.mpf(float(v))) for v in t]
    u = _ulps(got, ref)
    assert u.max() < 4.0, (u.max(), t[np.argmax(u)])
    assert np.abs(got - np.arcsin(t)).max() < 1e-15
    assert lean_asin(np.array([1.0 + 2e-16]))[0] == lean_asin(np.array([1.0]))[0]   # one ulp above 1: taken for 1
