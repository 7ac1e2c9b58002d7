/*
 * ref_basalt_probe.cpp -- C-ABI shim around the REFERENCE's own vendored Basalt So3Spline / Sophus.
 * TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code: it only #includes the headers where they lie under
 * /root/reference/thirdparty/basalt-headers (recipe: oracle/Makefile target `ref`; output
 * oracle/_ref/libbasalt_ref.so, git-ignored).  It is the live compiled reference used to
 *   (1) validate oracle/so3_spline.c, and
 *   (2) generate the golden vectors committed under tests/golden/ (oracle/gen_golden.py).
 * The calls mirror how the reference drives the spline:
 *   src/backend/trajectory.cpp:58-71   (ctor: dt_ns, start_ns, knotsPushBack)
 *   src/backend/trajectory.cpp:86-110 / :329-355 (evaluate + Jacobian blocks)
 *   src/backend/trajectory.cpp:236 / :497 (left-multiplicative update)
 *   src/backend/trajectory.cpp:176-178 / :448-450 (N.fullPivHouseholderQr().solve on the vendored Eigen)
 */
#include <cstdint>
#include <basalt/spline/so3_spline.h>

namespace {
template <int N>
int eval_n(int K, const double* knots, int64_t start_ns, int64_t dt_ns, int64_t t_ns, double* q_out, double* R,
           double* J, int* start_idx) {
  basalt::So3Spline<N> spline(dt_ns, start_ns);
  for (int i = 0; i < K; i++) {
    Eigen::Quaterniond q(knots[4 * i + 3], knots[4 * i], knots[4 * i + 1], knots[4 * i + 2]);
    Sophus::SO3d R0;
    R0.setQuaternion(q);
    spline.knotsPushBack(R0);
  }
  const int64_t st = t_ns - start_ns;
  if (st < 0 || st / dt_ns + N > K) return -1;  // the BASALT_ASSERTs would abort
  typename basalt::So3Spline<N>::JacobianStruct Js;
  Sophus::SO3d res = spline.evaluate(t_ns, J ? &Js : nullptr);
  const Eigen::Quaterniond& q = res.unit_quaternion();
  if (q_out) { q_out[0] = q.x(); q_out[1] = q.y(); q_out[2] = q.z(); q_out[3] = q.w(); }
  if (R) {
    Eigen::Matrix3d M = res.matrix();
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[3 * i + j] = M(i, j);
  }
  if (J) {
    if (start_idx) *start_idx = (int)Js.start_idx;
    for (int k = 0; k < N; k++)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) J[9 * k + 3 * i + j] = Js.d_val_d_knot[k](i, j);
  } else if (start_idx) {
    *start_idx = (int)(st / dt_ns);
  }
  return 0;
}
}  // namespace

extern "C" {
int ref_so3_spline_eval(int order, int K, const double* knots, int64_t start_ns, int64_t dt_ns, int64_t t_ns,
                        double* q_out, double* R, double* J, int* start_idx) {
  switch (order) {
    case 2: return eval_n<2>(K, knots, start_ns, dt_ns, t_ns, q_out, R, J, start_idx);
    case 4: return eval_n<4>(K, knots, start_ns, dt_ns, t_ns, q_out, R, J, start_idx);
    default: return -1;
  }
}
void ref_so3_exp(const double* w, double* q) {
  Sophus::SO3d r = Sophus::SO3d::exp(Eigen::Vector3d(w[0], w[1], w[2]));
  const Eigen::Quaterniond& u = r.unit_quaternion();
  q[0] = u.x(); q[1] = u.y(); q[2] = u.z(); q[3] = u.w();
}
void ref_so3_log(const double* q, double* w) {
  Sophus::SO3d r;
  r.setQuaternion(Eigen::Quaterniond(q[3], q[0], q[1], q[2]));
  Eigen::Vector3d v = r.log();
  w[0] = v[0]; w[1] = v[1]; w[2] = v[2];
}
void ref_so3_mul(const double* a, const double* b, double* out) {
  Sophus::SO3d ra, rb;
  ra.setQuaternion(Eigen::Quaterniond(a[3], a[0], a[1], a[2]));
  rb.setQuaternion(Eigen::Quaterniond(b[3], b[0], b[1], b[2]));
  const Sophus::SO3d r = ra * rb;
  const Eigen::Quaterniond& q = r.unit_quaternion();
  out[0] = q.x(); out[1] = q.y(); out[2] = q.z(); out[3] = q.w();
}
/* the solver call of src/backend/trajectory.cpp:176-178 / :448-450 on the vendored Eigen */
int ref_fullpiv_qr_solve(int rows, int cols, const double* A_rowmajor, const double* b, double* x) {
  Eigen::MatrixXd N(rows, cols);
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) N(i, j) = A_rowmajor[(size_t)i * cols + j];
  Eigen::VectorXd D(rows);
  for (int i = 0; i < rows; i++) D(i) = b[i];
  auto qr = N.fullPivHouseholderQr();
  Eigen::VectorXd P = qr.solve(D);
  for (int j = 0; j < cols; j++) x[j] = P(j);
  return (int)qr.rank();
}
void ref_so3_left_update(double* k, const double* drot) {
  Sophus::SO3d r;
  r.setQuaternion(Eigen::Quaterniond(k[3], k[0], k[1], k[2]));
  Sophus::SO3d u = Sophus::SO3d::exp(Eigen::Vector3d(drot[0], drot[1], drot[2])) * r;
  const Eigen::Quaterniond& q = u.unit_quaternion();
  k[0] = q.x(); k[1] = q.y(); k[2] = q.z(); k[3] = q.w();
}
}
