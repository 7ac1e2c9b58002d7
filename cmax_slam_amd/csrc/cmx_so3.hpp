// cmx_so3.hpp -- fp64 SO(3) / cumulative B-spline math shared by the pose-table kernel (device) and the
// C-ABI host code.  Written for HIP (gfx950); every function is __host__ __device__.
//
// Semantics follow what the reference evaluates per event batch:
//   Trajectory::evaluate -> basalt::So3Spline<N>::evaluate
//     (reference: src/backend/trajectory.cpp:86-110,329-355; thirdparty/basalt-headers/include/basalt/spline/
//      so3_spline.h:218-274; utils/sophus_utils.hpp:332-414; Sophus so3.hpp exp/log/product)
// i.e. unit quaternions (x,y,z,w), products re-normalised, atan-based log, left Jacobians, Jacobian of the
// value w.r.t. LEFT perturbations of the knots.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace cmx {

#define CMX_HD __host__ __device__ __forceinline__

struct Quat { double x, y, z, w; };
struct Mat3 { double m[9]; };

constexpr double kSophusEps = 1e-10;
constexpr double kPi = 3.141592653589793238462643383279502884;

CMX_HD Quat q_normalized(Quat q) {
  const double len = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / len, q.y / len, q.z / len, q.w / len};
}
CMX_HD Quat q_conj(Quat a) { return Quat{-a.x, -a.y, -a.z, a.w}; }
CMX_HD Quat q_mul(Quat a, Quat b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return q_normalized(r);
}
// product re-normalised with one reciprocal instead of four divisions (pose-table chain; <= 1 ulp from q_mul)
CMX_HD Quat q_mul_rcp(Quat a, Quat b) {
#pragma clang fp contract(fast)  // device pose-table chain only (the host has no FMA to contract into): see spline_eval_pre
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  const double inv = 1.0 / sqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  return Quat{r.x * inv, r.y * inv, r.z * inv, r.w * inv};
}
CMX_HD Quat so3_exp(double wx, double wy, double wz) {
  const double theta_sq = wx * wx + wy * wy + wz * wz;
  double imag, real;
  if (theta_sq < kSophusEps * kSophusEps) {
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
  } else {
    const double theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  return Quat{imag * wx, imag * wy, imag * wz, real};
}
CMX_HD void so3_log(Quat q, double out[3]) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  const double w = q.w;
  double f;
  if (n2 < kSophusEps * kSophusEps) {
    f = 2.0 / w - (2.0 / 3.0) * n2 / (w * (w * w));
  } else {
    const double n = sqrt(n2);
    if (fabs(w) < kSophusEps) f = (w > 0 ? kPi : -kPi) / n;
    else f = 2.0 * atan(n / w) / n;
  }
  out[0] = f * q.x; out[1] = f * q.y; out[2] = f * q.z;
}
CMX_HD Mat3 q_to_R(Quat q) {
#pragma clang fp contract(fast)
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}
CMX_HD Mat3 m3_mul(const Mat3 &a, const Mat3 &b) {
#pragma clang fp contract(fast)  // 9 multiplies + 18 FMAs instead of 27 + 18 on the device (the pose table is ~15 of these per batch)
  Mat3 o;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      o.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return o;
}
CMX_HD Mat3 m3_hat(const double p[3]) {
  Mat3 H;
  H.m[0] = 0; H.m[1] = -p[2]; H.m[2] = p[1];
  H.m[3] = p[2]; H.m[4] = 0; H.m[5] = -p[0];
  H.m[6] = -p[1]; H.m[7] = p[0]; H.m[8] = 0;
  return H;
}
CMX_HD Mat3 left_jacobian(const double phi[3]) {
  const double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const Mat3 H = m3_hat(phi), H2 = m3_mul(H, H);
  Mat3 J;
#pragma unroll
  for (int i = 0; i < 9; i++) J.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (n2 > kSophusEps) {
    const double n = sqrt(n2), n3 = n2 * n;
    const double c = cos(n), s = sin(n);
#pragma unroll
    for (int i = 0; i < 9; i++) J.m[i] += H.m[i] * (1 - c) / n2;
#pragma unroll
    for (int i = 0; i < 9; i++) J.m[i] += H2.m[i] * (n - s) / n3;
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++) J.m[i] += H.m[i] / 2;
#pragma unroll
    for (int i = 0; i < 9; i++) J.m[i] += H2.m[i] / 6;
  }
  return J;
}
CMX_HD Mat3 left_jacobian_inv(const double phi[3]) {
  const double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const Mat3 H = m3_hat(phi), H2 = m3_mul(H, H);
  Mat3 J;
#pragma unroll
  for (int i = 0; i < 9; i++) J.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 9; i++) J.m[i] -= H.m[i] / 2;
  if (n2 > kSophusEps) {
    const double n = sqrt(n2);
    if (n < kPi - 1e-5 /* sqrt(1e-10) */) {
      const double f = 1 / n2 - (1 + cos(n)) / (2 * n * sin(n));
#pragma unroll
      for (int i = 0; i < 9; i++) J.m[i] += H2.m[i] * f;
    } else {
#pragma unroll
      for (int i = 0; i < 9; i++) J.m[i] += H2.m[i] / (kPi * kPi);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++) J.m[i] += H2.m[i] / 12;
  }
  return J;
}

// Spline description passed by value to the pose-table kernel (K <= kMaxKnots knots of the temp trajectory).
constexpr int kMaxKnots = 64;
constexpr int kMaxOrder = 4;
struct SplineArgs {
  int order;      // 2 or 4
  int K;
  long long start_ns, dt_ns;
  double blend[kMaxOrder * kMaxOrder];  // cumulative blending matrix, row-major (host: blending_matrix())
  Quat knots[kMaxKnots];
};

// value R (row-major) and, if WANT_J, the `order` 3x3 blocks d_val_d_knot[i] (row-major, fp64)
template <int N, bool WANT_J>
CMX_HD void spline_eval(const SplineArgs &sp, long long t_ns, Mat3 &R, Mat3 *Jblocks, int &start_idx,
                        Quat *q_out = nullptr) {
  const long long st = t_ns - sp.start_ns;
  const long long s = st / sp.dt_ns;
  const double u = (double)(st % sp.dt_ns) / (double)sp.dt_ns;
  double p[N], coeff[N];
  p[0] = 1.0;
  double ti = u;
#pragma unroll
  for (int j = 1; j < N; j++) { p[j] = 1.0 * ti; ti = ti * u; }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double a = 0;
#pragma unroll
    for (int j = 0; j < N; j++) a += sp.blend[i * N + j] * p[j];
    coeff[i] = a;
  }
  start_idx = (int)s;
  Quat res = sp.knots[s];
  Mat3 Jh;
#pragma unroll
  for (int i = 0; i < 9; i++) Jh.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    const Quat p0 = sp.knots[s + i], p1 = sp.knots[s + i + 1];
    const Quat r01 = q_mul(q_conj(p0), p1);
    double delta[3], kdelta[3];
    so3_log(r01, delta);
#pragma unroll
    for (int c = 0; c < 3; c++) kdelta[c] = delta[c] * coeff[i + 1];
    if (WANT_J) {
      const Mat3 Jinv = left_jacobian_inv(delta);
      const Mat3 Jk = left_jacobian(kdelta);
      Jblocks[i] = Jh;
      Mat3 T = q_to_R(res);
#pragma unroll
      for (int c = 0; c < 9; c++) T.m[c] = coeff[i + 1] * T.m[c];
      T = m3_mul(T, Jk);
      T = m3_mul(T, Jinv);
      Jh = m3_mul(T, q_to_R(q_conj(p0)));
#pragma unroll
      for (int c = 0; c < 9; c++) Jblocks[i].m[c] -= Jh.m[c];
    }
    res = q_mul(res, so3_exp(kdelta[0], kdelta[1], kdelta[2]));
  }
  if (WANT_J) Jblocks[N - 1] = Jh;
  R = q_to_R(res);
  if (q_out) *q_out = res;
}

// ---- the same evaluation with everything that depends on the KNOTS only taken out of the per-batch chain.
// Of the work spline_eval does per segment, log(knot_i^-1 knot_{i+1}), its inverse left Jacobian and R(knot_i^-1) are
// the same for every batch of an evaluation: the host computes them once per evaluation (K-1 pairs, < 1 us) and they
// travel with the knots as kernel arguments.  The pose-table kernel is one thread per batch on < 1 wave per SIMD, i.e.
// bound by the length of its dependent fp64 chain: per segment this leaves one sqrt, one sin/cos pair (shared by
// exp(k delta) and the left Jacobian, which only feeds the fp32 Jacobian table) and two scalar divisions.
constexpr int kMaxKnotsPre = 16;
struct PairConsts {
  double delta[3];  // log(knot_i^-1 * knot_{i+1})
  double Jinv[9];   // leftJacobianInvSO3(delta)
  double Rc[9];     // R(knot_i^-1)
};
struct SplineArgsPre {
  int order, K;
  long long start_ns, dt_ns;
  double blend[kMaxOrder * kMaxOrder];
  Quat knots[kMaxKnotsPre];
  PairConsts pair[kMaxKnotsPre - 1];
};
static_assert(sizeof(SplineArgsPre) <= 3600, "SplineArgsPre travels as a kernel argument (4 KB limit with the others)");

inline void spline_precompute(const SplineArgs &sp, SplineArgsPre &o) {  // host, once per evaluation
  o.order = sp.order; o.K = sp.K; o.start_ns = sp.start_ns; o.dt_ns = sp.dt_ns;
  for (int i = 0; i < kMaxOrder * kMaxOrder; i++) o.blend[i] = sp.blend[i];
  for (int i = 0; i < sp.K; i++) o.knots[i] = sp.knots[i];
  for (int i = 0; i + 1 < sp.K; i++) {
    const Quat p0 = sp.knots[i], p1 = sp.knots[i + 1];
    so3_log(q_mul(q_conj(p0), p1), o.pair[i].delta);
    const Mat3 Ji = left_jacobian_inv(o.pair[i].delta), Rc = q_to_R(q_conj(p0));
    for (int c = 0; c < 9; c++) { o.pair[i].Jinv[c] = Ji.m[c]; o.pair[i].Rc[c] = Rc.m[c]; }
  }
}

// knots / pair: where the knot and pair constants are read from -- sp's own arrays, or a copy (the device kernel stages
// them in LDS: indexed per lane in the kernel-argument segment they cost a dependent, uncached round trip)
template <int N, bool WANT_J>
CMX_HD void spline_eval_pre(const SplineArgsPre &sp, const Quat *knots, const PairConsts *pair, long long t_ns, Mat3 &R,
                            Mat3 *Jblocks, int &start_idx) {
  // Contracted on the device (round 5): one wave per SIMD walks ~1850 dependent fp64 operations per batch -- the kernel is that chain's
  // latency; FMAs shorten it.  Pinned against the reference's compiled Basalt at 2e-14 (tests/test_gpu_pose_table.py); an FMA rounds once
  // where mul + add round twice.
#pragma clang fp contract(fast)
  const long long st = t_ns - sp.start_ns;
  const long long s = st / sp.dt_ns;
  const double u = (double)(st % sp.dt_ns) / (double)sp.dt_ns;
  double p[N], coeff[N];
  p[0] = 1.0;
  double ti = u;
#pragma unroll
  for (int j = 1; j < N; j++) { p[j] = 1.0 * ti; ti = ti * u; }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double a = 0;
#pragma unroll
    for (int j = 0; j < N; j++) a += sp.blend[i * N + j] * p[j];
    coeff[i] = a;
  }
  start_idx = (int)s;
  Quat res = knots[s];
  Mat3 Jh;
#pragma unroll
  for (int i = 0; i < 9; i++) Jh.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    const PairConsts &pc = pair[s + i];
    const double k = coeff[i + 1];
    const double kd[3] = {pc.delta[0] * k, pc.delta[1] * k, pc.delta[2] * k};
    // exp(k delta): the operations of so3_exp, with sin / cos of the half angle kept for the Jacobian below
    const double theta_sq = kd[0] * kd[0] + kd[1] * kd[1] + kd[2] * kd[2];
    double imag, real, sh = 0, ch = 1, theta = 0;
    const bool tiny = theta_sq < kSophusEps * kSophusEps;
    if (tiny) {
      const double t4 = theta_sq * theta_sq;
      imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
      real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
      theta = sqrt(theta_sq);
      const double half = 0.5 * theta;
      sh = sin(half); ch = cos(half);
      imag = sh / theta;
      real = ch;
    }
    if (WANT_J) {
      // leftJacobianSO3(k delta) = I + a H + b H^2 with scalar a, b (fp32 consumers: one division each, and
      // sin(n) = 2 sh ch, 1 - cos(n) = 2 sh^2 from the half-angle pair)
      const Mat3 H = m3_hat(kd), H2 = m3_mul(H, H);
      double a, b;
      if (theta_sq > kSophusEps) {
        a = 2.0 * sh * sh / theta_sq;
        b = (theta - 2.0 * sh * ch) / (theta_sq * theta);
      } else {
        a = 0.5; b = 1.0 / 6.0;
      }
      Mat3 Jk;
#pragma unroll
      for (int c = 0; c < 9; c++) Jk.m[c] = ((c % 4 == 0) ? 1.0 : 0.0) + a * H.m[c] + b * H2.m[c];
      Jblocks[i] = Jh;
      Mat3 T = q_to_R(res), Ji, Rc;
#pragma unroll
      for (int c = 0; c < 9; c++) { T.m[c] = k * T.m[c]; Ji.m[c] = pc.Jinv[c]; Rc.m[c] = pc.Rc[c]; }
      T = m3_mul(T, Jk);
      T = m3_mul(T, Ji);
      Jh = m3_mul(T, Rc);
#pragma unroll
      for (int c = 0; c < 9; c++) Jblocks[i].m[c] -= Jh.m[c];
    }
    res = q_mul_rcp(res, Quat{imag * kd[0], imag * kd[1], imag * kd[2], real});
  }
  if (WANT_J) Jblocks[N - 1] = Jh;
  R = q_to_R(res);
}

// cumulative blending matrix of a uniform B-spline of order N (host only; tiny)
inline void blending_matrix(int N, double *m) {
  auto binom = [](int n, int k) {
    if (k > n) return 0.0;
    double r = 1;
    for (int d = 1; d <= k; ++d) { r *= (double)(n - (d - 1)); r /= (double)d; }
    return r;
  };
  for (int i = 0; i < N * N; i++) m[i] = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double sum = 0;
      for (int s = j; s < N; ++s) sum += pow(-1.0, s - j) * binom(N, s - j) * pow(N - s - 1.0, N - 1.0 - i);
      m[j * N + i] = binom(N - 1, N - 1 - i) * sum;
    }
  for (int i = 0; i < N; i++)
    for (int j = i + 1; j < N; j++)
      for (int c = 0; c < N; c++) m[i * N + c] += m[j * N + c];
  double factorial = 1;
  for (int i = 2; i < N; ++i) factorial *= i;
  for (int i = 0; i < N * N; i++) m[i] /= factorial;
}

}  // namespace cmx
