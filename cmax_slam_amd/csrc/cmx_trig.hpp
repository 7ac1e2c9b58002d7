// cmx_trig.hpp -- fp64 atan2 / asin for the back end's equirectangular projection (device only).
//
// ocml's atan2 + asin were 138 of the back-end splat's ~180 VALU instructions per event, and as many of the gather's ~320
// (62 + 76: asin carries a double-double correction for < 1 ulp).  The pixel coordinate needs nothing like
// that: these are plain polynomial forms, 1-3 ulp (4e-13 of a pixel at 4096 x 2048), ~85 instructions for the pair:
//   atan(a) = a + a r Qa(r),  r = a^2 in [0, 1],  a = min(|y|,|x|) / max(|y|,|x|) (reciprocal + Newton + one residual step)
//   asin(s) = s + s r Qs(r),  r = s^2 in [0, 1/4]; |t| > 1/2: asin(t) = pi/2 - 2 asin(sqrt((1 - |t|) / 2)), the square root by
//             v_rsq_f64 + Newton, skipped wave-uniformly when no lane needs it
// Coefficients: near-minimax fits in 60-digit arithmetic, tools/trig/fit_trig.py (atan: degree 20 in r, max error 1.16 ulp (mean 0.27); asin: degree 12 in r, max error 0.60 ulp (mean 0.25)).
// tests/test_trig_poly.py re-evaluates the same Horner forms in numpy against mpmath; the end-to-end check is the parity suite.
// Differences from ocml / glibc are of the order the two already differ by.
#pragma once
#include <hip/hip_runtime.h>

namespace cmx {

__device__ constexpr double kAtanQ[21] = {
    -3.33333333333333315e-01, 1.99999999999995542e-01, -1.42857142856484043e-01, 1.11111111072347987e-01,
    -9.09090896955740274e-02, 7.69230535467865517e-02, -6.66663643577769249e-02, 5.88207495637129429e-02,
    -5.26126573570945416e-02, 4.75208677357665601e-02, -4.30811965533047098e-02, 3.87264021404263234e-02,
    -3.37501320016197343e-02, 2.75679422973446782e-02, -2.02387069863935141e-02, 1.27561729076942978e-02,
    -6.57568334415169903e-03, 2.62297789190608921e-03, -7.51847252697382157e-04, 1.36872485314812196e-04,
    -1.18325054175556920e-05};
__device__ constexpr double kAsinQ[13] = {
    1.66666666666666685e-01, 7.49999999999843292e-02, 4.46428571463554288e-02, 3.03819441385312465e-02,
    2.23721729421498886e-02, 1.73523927208699726e-02, 1.39712129735529329e-02, 1.14791774151849057e-02,
    1.03228143501857793e-02, 5.45750671864035815e-03, 1.74008794426940214e-02, -1.48518870712472037e-02,
    2.87578513674215663e-02};

// One Horner step p <- p * r + c as the THREE-ADDRESS v_fma_f64.  Written as asm because the compiler turns the plain expression into
// v_mov_b64 tmp, c ; v_fmac_f64 tmp, p, r (the two-address form clobbers its addend, and the coefficient -- hoisted into a VGPR pair
// for the whole event loop -- must survive): 33 extra 64-bit moves per event in front of 33 FMAs, a tenth of the back-end kernels'
// VALU instructions (profiles/r05_be_valu.txt).  Same operation, same operands, same bits.
#ifndef CMX_HORNER_ASM
#define CMX_HORNER_ASM 1
#endif
__device__ __forceinline__ double horner_step(double p, double r, double c) {
#if CMX_HORNER_ASM
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "v"(c));
  return d;
#else
  return __builtin_fma(p, r, c);
#endif
}

__device__ __forceinline__ double trig_rcp(double d) {  // 1 / d, ~correctly rounded for normal d
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}

// atan2(y, x) for finite arguments; (0, 0) -> 0 like the C library
__device__ __forceinline__ double lean_atan2(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double hi = __builtin_fmax(ax, ay), lo = __builtin_fmin(ax, ay);
  const double inv = trig_rcp(hi);
  double a = lo * inv;
  a = __builtin_fma(__builtin_fma(-hi, a, lo), inv, a);  // residual step: a = lo / hi to the last place
  if (hi < 1e-290) a = hi > 0.0 ? lo / hi : 0.0;  // (a ray along the pole: the reciprocal of a subnormal overflows; never in practice)
  const double r = a * a;
  double p = kAtanQ[20];
#pragma unroll
  for (int k = 19; k >= 0; k--) p = horner_step(p, r, kAtanQ[k]);
  double t = __builtin_fma(a * r, p, a);
  constexpr double kPi = 3.14159265358979323846, kHalfPi = 1.57079632679489661923;
  t = ay > ax ? kHalfPi - t : t;
  t = x < 0.0 ? kPi - t : t;
  return __builtin_copysign(t, y);
}

// asin(t), |t| <= 1 (a |t| one ulp above 1 is taken for 1)
__device__ __forceinline__ double lean_asin(double t) {
  const double at = __builtin_fmin(__builtin_fabs(t), 1.0);
  const bool big = at > 0.5;
  double r = at * at, s = at;
  if (__any(big)) {  // wave-uniform: a window whose pitch stays below 30 degrees never takes the square root
    const double rb = __builtin_fma(at, -0.5, 0.5);
    double y = __builtin_amdgcn_rsq(rb);
    y = y * __builtin_fma(-0.5 * rb, y * y, 1.5);
    y = y * __builtin_fma(-0.5 * rb, y * y, 1.5);
    double sb = rb * y;
    sb = __builtin_fma(__builtin_fma(-sb, sb, rb), 0.5 * y, sb);  // residual step
    sb = rb > 0.0 ? sb : 0.0;
    r = big ? rb : r;
    s = big ? sb : s;
  }
  double p = kAsinQ[12];
#pragma unroll
  for (int k = 11; k >= 0; k--) p = horner_step(p, r, kAsinQ[k]);
  const double u = __builtin_fma(s * r, p, s);
  constexpr double kHalfPi = 1.57079632679489661923;
  const double v = big ? __builtin_fma(-2.0, u, kHalfPi) : u;
  return __builtin_copysign(v, t);
}

}  // namespace cmx
