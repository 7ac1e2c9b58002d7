/*
 * so3_spline.c -- cumulative uniform B-spline on SO(3) with Jacobians w.r.t. knots, restated.
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).
 *
 * Follows (all under thirdparty/basalt-headers/):
 *   include/basalt/spline/so3_spline.h:218-274   So3Spline<N>::evaluate
 *   include/basalt/spline/so3_spline.h:754-773   baseCoeffsWithTime<0>
 *   include/basalt/spline/spline_common.h:69-135 computeBlendingMatrix<N,double,true>, computeBaseCoefficients
 *   include/basalt/utils/sophus_utils.hpp:332-414 leftJacobianSO3, leftJacobianInvSO3
 *   thirdparty/Sophus/sophus/so3.hpp:229-231 (inverse), :247-291 (logAndTheta), :297-303 (normalize),
 *                                    :325-339 (product), :583-618 (expAndTheta); common.hpp:94 (epsilon=1e-10)
 *   Eigen QuaternionBase::toRotationMatrix
 * PINNED: validated against the vendored Basalt itself compiled from the reference tree
 * (oracle/_ref/libbasalt_ref.so) and against tests/golden/spline_*.npz produced by it.
 * Not restated: the association order inside Eigen's fixed-size 3x3 products (<= 1 ulp fp64).
 */
#include "cmax_oracle.h"
#include <math.h>
#include <string.h>

#define SOPHUS_EPS 1e-10
#define SOPHUS_PI 3.141592653589793238462643383279502884

typedef struct { double x, y, z, w; } quat;

/* SO3::normalize  so3.hpp:297-303 */
static quat q_normalize(quat q) {
  const double len = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= len; q.y /= len; q.z /= len; q.w /= len;
  return q;
}
/* SO3 * SO3  so3.hpp:325-339; the product type's constructor re-normalises (:480-486) */
static quat q_mul(quat a, quat b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return q_normalize(r);
}
static quat q_conj(quat a) { quat r = {-a.x, -a.y, -a.z, a.w}; return r; }

/* SO3::expAndTheta  so3.hpp:583-618 */
static quat so3_exp(const double w[3]) {
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (theta_sq < SOPHUS_EPS * SOPHUS_EPS) {
    const double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    const double theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  quat q = {imag * w[0], imag * w[1], imag * w[2], real};
  return q;
}
/* SO3::logAndTheta  so3.hpp:247-291 */
static void so3_log(quat q, double out[3]) {
  const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
  const double w = q.w;
  double two_atan_nbyw_by_n;
  if (squared_n < SOPHUS_EPS * SOPHUS_EPS) {
    const double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * squared_n / (w * squared_w);
  } else {
    const double n = sqrt(squared_n);
    if (fabs(w) < SOPHUS_EPS) {
      two_atan_nbyw_by_n = (w > 0 ? SOPHUS_PI : -SOPHUS_PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * atan(n / w) / n;
    }
  }
  out[0] = two_atan_nbyw_by_n * q.x;
  out[1] = two_atan_nbyw_by_n * q.y;
  out[2] = two_atan_nbyw_by_n * q.z;
}
/* Eigen::QuaternionBase::toRotationMatrix (row-major out) */
static void q_to_R(quat q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void m3_mul(const double *a, const double *b, double *o) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
  memcpy(o, t, sizeof(t));
}
static void hat(const double p[3], double H[9]) {
  H[0] = 0; H[1] = -p[2]; H[2] = p[1];
  H[3] = p[2]; H[4] = 0; H[5] = -p[0];
  H[6] = -p[1]; H[7] = p[0]; H[8] = 0;
}
/* Sophus::leftJacobianSO3  sophus_utils.hpp:332-362 */
static void left_jacobian(const double phi[3], double J[9]) {
  const double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double H[9], H2[9];
  hat(phi, H);
  m3_mul(H, H, H2);
  for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (n2 > SOPHUS_EPS) {
    const double n = sqrt(n2), n3 = n2 * n;
    for (int i = 0; i < 9; i++) J[i] += H[i] * (1 - cos(n)) / n2;
    for (int i = 0; i < 9; i++) J[i] += H2[i] * (n - sin(n)) / n3;
  } else {
    for (int i = 0; i < 9; i++) J[i] += H[i] / 2;
    for (int i = 0; i < 9; i++) J[i] += H2[i] / 6;
  }
}
/* Sophus::leftJacobianInvSO3  sophus_utils.hpp:372-414 */
static void left_jacobian_inv(const double phi[3], double J[9]) {
  const double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double H[9], H2[9];
  hat(phi, H);
  m3_mul(H, H, H2);
  for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 9; i++) J[i] -= H[i] / 2;
  if (n2 > SOPHUS_EPS) {
    const double n = sqrt(n2);
    if (n < M_PI - sqrt(SOPHUS_EPS)) {
      const double f = 1 / n2 - (1 + cos(n)) / (2 * n * sin(n));
      for (int i = 0; i < 9; i++) J[i] += H2[i] * f;
    } else {
      for (int i = 0; i < 9; i++) J[i] += H2[i] / (M_PI * M_PI);
    }
  } else {
    for (int i = 0; i < 9; i++) J[i] += H2[i] / 12;
  }
}

static double binom(int n, int k) { /* binomialCoefficient  spline_common.h:51-61 */
  if (k > n) return 0;
  double r = 1;
  for (int d = 1; d <= k; ++d) { r *= (double)(n - (d - 1)); r /= (double)d; } /* exact: small integers */
  return r;
}
/* computeBlendingMatrix<N,double,true>  spline_common.h:69-100 (row-major m[row*N+col]) */
static void blending_matrix(int N, double *m) {
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

void orc_so3_exp(const double w[3], double q[4]) { quat r = so3_exp(w); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
void orc_so3_log(const double q[4], double w[3]) { quat r = {q[0], q[1], q[2], q[3]}; so3_log(r, w); }

/* SO3 * SO3 (re-normalised), so3.hpp:325-339 */
void orc_so3_mul(const double a[4], const double b[4], double out[4]) {
  quat qa = {a[0], a[1], a[2], a[3]}, qb = {b[0], b[1], b[2], b[3]};
  quat r = q_mul(qa, qb);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

/* spline_.getKnot(i) = SO3::exp(drot) * spline_.getKnot(i)   trajectory.cpp:236 / :497 */
void orc_so3_left_update(double *k, const double drot[3]) {
  quat q = {k[0], k[1], k[2], k[3]};
  quat r = q_mul(so3_exp(drot), q);
  k[0] = r.x; k[1] = r.y; k[2] = r.z; k[3] = r.w;
}

/* start of the temp trajectory: int64_t(1e9 * (t_beg_ + idx_traj_beg * dt_knots_))
 * trajectory.cpp:255-256 -> ctor :58-67 / :302-311 */
int64_t orc_traj_temp_start_ns(double t_beg, int idx_traj_beg, double dt_knots) {
  const double t = t_beg + idx_traj_beg * dt_knots;
  return (int64_t)(1e9 * t);
}

/* So3Spline<N>::evaluate  so3_spline.h:218-274 */
int orc_so3_spline_eval(int N, int K, const double *knots, int64_t start_ns, int64_t dt_ns, int64_t t_ns,
                        double *q_out, double *R_out, double *J, int *start_idx) {
  const int DEG = N - 1;
  if (N < 2 || N > 6) return -1;
  const int64_t st_ns = t_ns - start_ns;
  if (st_ns < 0) return -1; /* BASALT_ASSERT :221 */
  const int64_t s = st_ns / dt_ns;
  const double u = (double)(st_ns % dt_ns) / (double)dt_ns;
  if (s < 0 || (int64_t)(s + N) > (int64_t)K) return -1; /* :227-230 */

  /* baseCoeffsWithTime<0>: p = [1, u, u^2, ...] (BASE_COEFFICIENTS row 0 is all ones) :754-773 */
  double p[6], coeff[6], M[36];
  p[0] = 1.0;
  double ti = u;
  for (int j = 1; j < N; j++) { p[j] = 1.0 * ti; ti = ti * u; }
  blending_matrix(N, M);
  for (int i = 0; i < N; i++) {
    double a = 0;
    for (int j = 0; j < N; j++) a += M[i * N + j] * p[j];
    coeff[i] = a;
  }

  quat res = {knots[4 * s], knots[4 * s + 1], knots[4 * s + 2], knots[4 * s + 3]};
  double J_helper[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (start_idx) *start_idx = (int)s;

  for (int i = 0; i < DEG; i++) {
    const double *k0 = knots + 4 * (s + i), *k1 = knots + 4 * (s + i + 1);
    const quat p0 = {k0[0], k0[1], k0[2], k0[3]}, p1 = {k1[0], k1[1], k1[2], k1[3]};
    const quat r01 = q_mul(q_conj(p0), p1);
    double delta[3], kdelta[3];
    so3_log(r01, delta);
    for (int c = 0; c < 3; c++) kdelta[c] = delta[c] * coeff[i + 1];
    if (J) {
      double Jl_inv_delta[9], Jl_k_delta[9], Rres[9], Rp0inv[9], T[9];
      left_jacobian_inv(delta, Jl_inv_delta);
      left_jacobian(kdelta, Jl_k_delta);
      memcpy(J + 9 * i, J_helper, sizeof(J_helper)); /* d_val_d_knot[i] = J_helper */
      q_to_R(res, Rres);
      q_to_R(q_conj(p0), Rp0inv);
      /* J_helper = coeff[i+1] * res.matrix() * Jl_k_delta * Jl_inv_delta * p0.inverse().matrix() */
      for (int c = 0; c < 9; c++) T[c] = coeff[i + 1] * Rres[c];
      m3_mul(T, Jl_k_delta, T);
      m3_mul(T, Jl_inv_delta, T);
      m3_mul(T, Rp0inv, J_helper);
      for (int c = 0; c < 9; c++) J[9 * i + c] -= J_helper[c];
    }
    res = q_mul(res, so3_exp(kdelta)); /* res *= SO3::exp(kdelta) */
  }
  if (J) memcpy(J + 9 * DEG, J_helper, sizeof(J_helper));
  if (q_out) { q_out[0] = res.x; q_out[1] = res.y; q_out[2] = res.z; q_out[3] = res.w; }
  if (R_out) q_to_R(res, R_out);
  return 0;
}
