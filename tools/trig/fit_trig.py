"""Coefficients of the back end's fp64 atan / asin polynomials (cmx_trig.hpp), and their measured accuracy.

  atan(a) = a + a r Qa(r),  r = a^2 in [0, 1]      (a = min(|x|,|z|) / max(|x|,|z|); octant fix-up outside)
  asin(s) = s + s r Qs(r),  r = s^2 in [0, 1/4]    (|t| > 1/2: asin(t) = pi/2 - 2 asin(sqrt((1 - t) / 2)))

Near-minimax fits (Chebyshev interpolation in 60-digit arithmetic, mpmath), coefficients rounded to double; the error reported is
that of the fp64 Horner evaluation (numpy) against the 60-digit value on a dense grid, in units of the result's last place.
Usage: python tools/trig/fit_trig.py [deg_atan deg_asin]   -> prints the C arrays and the errors"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def q_atan(r):
    r = mp.mpf(r)
    if r < mp.mpf(10) ** -12:  # series: -1/3 + r/5 - r^2/7
        return -mp.mpf(1) / 3 + r / 5 - r * r / 7
    s = mp.sqrt(r)
    return (mp.atan(s) / s - 1) / r


def q_asin(r):
    r = mp.mpf(r)
    if r < mp.mpf(10) ** -12:  # series: 1/6 + 3 r/40 + 15 r^2/336
        return mp.mpf(1) / 6 + 3 * r / 40 + 15 * r * r / 336
    s = mp.sqrt(r)
    return (mp.asin(s) / s - 1) / r


def fit(f, lo, hi, deg):
    c = mp.chebyfit(f, [lo, hi], deg + 1)   # deg+1 coefficients, highest power first
    return [float(v) for v in c][::-1]      # lowest power first, rounded to double


def horner(c, r):
    p = np.full_like(r, c[-1])
    for k in range(len(c) - 2, -1, -1):
        p = p * r + c[k]                     # (the device code uses fma: at least as accurate)
    return p


def ulp_err(approx, exact_mp):
    exact = np.array([float(v) for v in exact_mp])
    err = np.array([float(abs(mp.mpf(float(a)) - e)) for a, e in zip(approx, exact_mp)])
    return err / np.spacing(np.abs(exact))


def main():
    da = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    ds = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    ca = fit(q_atan, 0, 1, da)
    cs = fit(q_asin, 0, mp.mpf(1) / 4, ds)
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.random(20000), np.linspace(0, 1, 2001), 1 - 10.0 ** -rng.uniform(1, 15, 2000)])
    a = a[(a >= 0) & (a <= 1)]
    r = a * a
    ya = a + a * r * horner(ca, r)
    ea = ulp_err(ya, [mp.atan(mp.mpf(float(v))) for v in a])
    s = np.concatenate([0.5 * rng.random(20000), np.linspace(0, 0.5, 2001)])
    r = s * s
    ys = s + s * r * horner(cs, r)
    es = ulp_err(ys, [mp.asin(mp.mpf(float(v))) for v in s])
    print("atan: degree %d in r, max error %.2f ulp (mean %.2f); asin: degree %d in r, max error %.2f ulp (mean %.2f)"
          % (da, ea.max(), ea.mean(), ds, es.max(), es.mean()))
    for name, c in (("kAtanQ", ca), ("kAsinQ", cs)):
        print("constexpr double %s[%d] = {%s};" % (name, len(c), ", ".join("%.17e" % v for v in c)))


if __name__ == "__main__":
    main()
