// cmx_trajinit.cpp -- host-side control-pose initialisation and bearing-vector table:
// the small amount of fp64 linear algebra that turns the front end's angular velocities into the control poses a
// back-end window starts from, and the once-per-camera bearing LUT.  Dozens of poses x a handful of control poses
// per window: this is host work by nature (the reference does it on the back-end thread between two solves), so
// it lives above the device path as plain C++ behind the same C ABI.  No device calls in this file.
//
//   cmx_integrate_ang_vel         PoseGraphOptimizer::integrateAngVel      src/backend/pose_graph_optimizer.cpp:191-222
//   cmx_num_ctrl_poses            {Linear,Cubic}Trajectory::generateCtrlPoses   src/backend/trajectory.cpp:205-214, :480-489
//   cmx_fit_ctrl_poses            {Linear,Cubic}Trajectory::fitCtrlPoses   src/backend/trajectory.cpp:112-192, :357-464
//   cmx_traj_incremental_update   {Linear,Cubic}Trajectory::incrementalUpdate   src/backend/trajectory.cpp:221-238, :491-499
//   cmx_traj_evaluate             {Linear,Cubic}Trajectory::evaluate (value only)  src/backend/trajectory.cpp:86-110, :329-355
//   cmx_bearing_lut               CMaxSLAM::precomputeBearingVectors       src/cmax_slam.cpp:106-120
#include "../../include/cmax_hip.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "cmx_so3.hpp"

namespace {
using namespace cmx;

constexpr long long kNsPerSec = 1000000000LL;

// ros::Duration::toSec() of an exact nanosecond difference: seconds floor-normalised, nsec in [0, 1e9)
double duration_sec(long long d) {
  long long s = d / kNsPerSec, n = d % kNsPerSec;
  if (n < 0) { n += kNsPerSec; s -= 1; }
  return (double)s + 1e-9 * (double)n;
}
// ros::Time::toSec()
double time_sec(long long t) { return (double)(t / kNsPerSec) + 1e-9 * (double)(t % kNsPerSec); }

Quat load_q(const double *p) { return Quat{p[0], p[1], p[2], p[3]}; }
void store_q(double *p, Quat q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }

// Dense least squares by Householder QR with full pivoting, returning the same "basic" solution Eigen's
// FullPivHouseholderQR::solve yields (free variables of a rank-deficient system are set to zero, not minimum-norm):
// the reference calls N.fullPivHouseholderQr().solve(D) three times (trajectory.cpp:176-178); we factor once and
// push the three right-hand sides through together.
class FullPivQR {
 public:
  FullPivQR(int rows, int cols, std::vector<double> a_colmajor) : m_(rows), n_(cols), a_(std::move(a_colmajor)) {
    const int size = std::min(m_, n_);
    tau_.assign(size, 0.0);
    row_swap_.resize(size);
    col_swap_.resize(size);
    const double eps = 2.220446049250313e-16;
    const double negligible = eps * (double)size;
    double first_pivot_mag = 0, max_diag = 0;
    int nonzero = size;
    for (int k = 0; k < size; k++) {
      // pivot = largest magnitude of the trailing block, first one in column-major order on ties
      int pr = k, pc = k;
      double mag = -1;
      for (int j = k; j < n_; j++)
        for (int i = k; i < m_; i++)
          if (fabs(at(i, j)) > mag) { mag = fabs(at(i, j)); pr = i; pc = j; }
      if (k == 0) first_pivot_mag = mag;
      if (mag <= first_pivot_mag * negligible) {  // trailing block is numerically zero: rank found
        nonzero = k;
        for (int i = k; i < size; i++) { row_swap_[i] = i; col_swap_[i] = i; }
        break;
      }
      row_swap_[k] = pr;
      col_swap_[k] = pc;
      if (pr != k)
        for (int j = k; j < n_; j++) std::swap(at(k, j), at(pr, j));
      if (pc != k)
        for (int i = 0; i < m_; i++) std::swap(at(i, k), at(i, pc));
      // reflector that maps column k (rows k..) onto beta * e1; v = [1, essential], stored below the diagonal
      double tail2 = 0;
      for (int i = k + 1; i < m_; i++) tail2 += at(i, k) * at(i, k);
      const double head = at(k, k);
      double beta;
      if (tail2 <= 2.2250738585072014e-308) {
        beta = head;
        for (int i = k + 1; i < m_; i++) at(i, k) = 0;
      } else {
        beta = sqrt(head * head + tail2);
        if (head >= 0) beta = -beta;
        for (int i = k + 1; i < m_; i++) at(i, k) /= (head - beta);
        tau_[k] = (beta - head) / beta;
      }
      at(k, k) = beta;
      max_diag = std::max(max_diag, fabs(beta));
      for (int j = k + 1; j < n_; j++) reflect(k, &at(0, j));
    }
    perm_.resize(n_);
    for (int j = 0; j < n_; j++) perm_[j] = j;
    for (int k = 0; k < size; k++) std::swap(perm_[k], perm_[col_swap_[k]]);
    rank_ = 0;
    for (int i = 0; i < nonzero; i++) rank_ += fabs(at(i, i)) > max_diag * negligible;
  }
  int rank() const { return rank_; }
  // x (n_) from b (m_)
  void solve(const double *b, double *x) const {
    std::vector<double> c(b, b + m_);
    for (int k = 0; k < rank_; k++) {
      std::swap(c[k], c[row_swap_[k]]);
      reflect(k, c.data());
    }
    for (int i = rank_ - 1; i >= 0; i--) {
      double s = c[i];
      for (int j = i + 1; j < rank_; j++) s -= at(i, j) * c[j];
      c[i] = s / at(i, i);
    }
    for (int j = 0; j < n_; j++) x[j] = 0;
    for (int i = 0; i < rank_; i++) x[perm_[i]] = c[i];
  }

 private:
  double &at(int i, int j) { return a_[(size_t)j * m_ + i]; }
  double at(int i, int j) const { return a_[(size_t)j * m_ + i]; }
  // col (length m_, rows k.. are touched) <- (I - tau v v^T) col
  void reflect(int k, double *col) const {
    if (m_ - k == 1) { col[k] *= 1.0 - tau_[k]; return; }
    if (tau_[k] == 0) return;
    double dot = 0;
    for (int i = k + 1; i < m_; i++) dot += at(i, k) * col[i];
    dot += col[k];
    col[k] -= tau_[k] * dot;
    for (int i = k + 1; i < m_; i++) col[i] -= tau_[k] * at(i, k) * dot;
  }
  int m_, n_, rank_ = 0;
  std::vector<double> a_, tau_;
  std::vector<int> row_swap_, col_swap_, perm_;
};

template <int N>
int evaluate_quat(int K, const double *knots, long long start_ns, long long dt_ns, long long t_ns, double *out) {
  const long long st = t_ns - start_ns;
  if (st < 0 || st / dt_ns + N > K) return CMX_ERR_INVALID_ARG;  // Basalt would assert (so3_spline.h:221-230)
  std::vector<double> blend(N * N);
  blending_matrix(N, blend.data());
  const long long s = st / dt_ns;
  const double u = (double)(st % dt_ns) / (double)dt_ns;
  double p[N], coeff[N];
  p[0] = 1.0;
  double ti = u;
  for (int j = 1; j < N; j++) { p[j] = 1.0 * ti; ti = ti * u; }
  for (int i = 0; i < N; i++) {
    double a = 0;
    for (int j = 0; j < N; j++) a += blend[i * N + j] * p[j];
    coeff[i] = a;
  }
  Quat res = load_q(knots + 4 * s);
  for (int i = 0; i < N - 1; i++) {
    const Quat p0 = load_q(knots + 4 * (s + i)), p1 = load_q(knots + 4 * (s + i + 1));
    double delta[3];
    so3_log(q_mul(q_conj(p0), p1), delta);
    res = q_mul(res, so3_exp(delta[0] * coeff[i + 1], delta[1] * coeff[i + 1], delta[2] * coeff[i + 1]));
  }
  store_q(out, res);
  return CMX_OK;
}
}  // namespace

extern "C" {

int cmx_integrate_ang_vel(int n, const int64_t *t_ns, const double *ang_vel, int64_t pose_t_ns, const double pose_quat[4],
                          int64_t *prev_t_ns, double prev_ang_vel[3], int first_time_window, int64_t *out_t_ns,
                          double *out_quat, int *n_out) {
  if (n < 0 || !pose_quat || !prev_t_ns || !prev_ang_vel || !n_out || (n > 0 && (!t_ns || !ang_vel || !out_t_ns || !out_quat)))
    return CMX_ERR_INVALID_ARG;
  for (int i = 1; i < n; i++)
    if (t_ns[i] <= t_ns[i - 1]) return CMX_ERR_TIME_ORDER;  // the reference holds them in a std::map keyed by stamp
  long long stamp = pose_t_ns;
  Quat pose = load_q(pose_quat);
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (!(t_ns[i] > *prev_t_ns) && !first_time_window) continue;  // "Wrong ang_vel timestamp, skip"
    const double dt = duration_sec((long long)t_ns[i] - stamp);
    const double *w = ang_vel + 3 * i;
    const double rx = dt * ((prev_ang_vel[0] + w[0]) / 2.0), ry = dt * ((prev_ang_vel[1] + w[1]) / 2.0),
                 rz = dt * ((prev_ang_vel[2] + w[2]) / 2.0);
    stamp = t_ns[i];
    pose = q_mul(pose, so3_exp(rx, ry, rz));  // trapezoidal increment, post-multiplied
    out_t_ns[m] = stamp;
    store_q(out_quat + 4 * m, pose);
    m++;
    *prev_t_ns = t_ns[i];
    prev_ang_vel[0] = w[0]; prev_ang_vel[1] = w[1]; prev_ang_vel[2] = w[2];
  }
  *n_out = m;
  return CMX_OK;
}

int cmx_num_ctrl_poses(int order, int64_t t_beg_ns, int64_t t_end_ns, double dt_knots) {
  if ((order != 2 && order != 4) || !(dt_knots > 0)) return -1;
  return (int)round(duration_sec((long long)t_end_ns - (long long)t_beg_ns) / dt_knots) + (order == 4 ? 3 : 1);
}

int cmx_fit_ctrl_poses(int order, int n_poses, const int64_t *t_ns, const double *quat, double t_beg_sec, double dt_knots,
                       int num_cps, double *out_quat) {
  if ((order != 2 && order != 4) || !t_ns || !quat || !out_quat || !(dt_knots > 0)) return CMX_ERR_INVALID_ARG;
  if (num_cps < order || n_poses < num_cps) return CMX_ERR_INVALID_ARG;  // CHECK_GE(poses.size(), num_cps)
  // uniform B-spline basis, rows = powers of u (1, u, u^2, u^3), columns = the `order` supporting control poses
  static const double kBasis2[4] = {1.0, 0.0, -1.0, 1.0};
  static const double kBasis4[16] = {1. / 6, 2. / 3, 1. / 6, 0.0, -0.5, 0.0, 0.5, 0.0,
                                     0.5,    -1.0,   0.5,    0.0, -1. / 6, 0.5, -0.5, 1. / 6};
  const double *basis = order == 2 ? kBasis2 : kBasis4;
  const Quat offset = load_q(quat);
  const Quat offset_inv = q_conj(offset);
  std::vector<double> design((size_t)n_poses * num_cps, 0.0);  // column-major n_poses x num_cps
  std::vector<double> rhs((size_t)3 * n_poses);
  for (int p = 0; p < n_poses; p++) {
    const double t = time_sec(t_ns[p]);
    const int seg = (int)floor((t - t_beg_sec) / dt_knots);
    if (seg < 0 || seg + order > num_cps) return CMX_ERR_INVALID_ARG;  // pose outside the span of the new control poses
    const double u = (t - (seg * dt_knots + t_beg_sec)) / dt_knots;
    double upow[4];
    for (int i = 0; i < order; i++) upow[i] = pow(u, i);
    for (int j = 0; j < order; j++) {
      double s = 0;
      for (int i = 0; i < order; i++) s += upow[i] * basis[i * order + j];
      design[(size_t)(seg + j) * n_poses + p] = s;
    }
    double w[3];
    so3_log(q_mul(offset_inv, load_q(quat + 4 * p)), w);  // lift: increment w.r.t. the first pose
    rhs[p] = w[0]; rhs[n_poses + p] = w[1]; rhs[2 * (size_t)n_poses + p] = w[2];
  }
  const FullPivQR qr(n_poses, num_cps, std::move(design));
  std::vector<double> sol((size_t)3 * num_cps);
  for (int a = 0; a < 3; a++) qr.solve(rhs.data() + (size_t)a * n_poses, sol.data() + (size_t)a * num_cps);
  for (int i = 0; i < num_cps; i++)  // retract
    store_q(out_quat + 4 * i, q_mul(offset, so3_exp(sol[i], sol[num_cps + i], sol[2 * (size_t)num_cps + i])));
  return CMX_OK;
}

int cmx_traj_incremental_update(int K, double *knots, int idx_beg, int n_params, const double *drotv) {
  if (K < 0 || idx_beg < 0 || !knots || (n_params > 0 && !drotv)) return CMX_ERR_INVALID_ARG;
  if (n_params % 3 != 0 || idx_beg + n_params / 3 != K) return CMX_ERR_INVALID_ARG;  // CHECK_EQ(idx_beg + drotv.size(), size())
  for (int i = idx_beg; i < K; i++) {
    const double *d = drotv + 3 * (i - idx_beg);
    store_q(knots + 4 * i, q_mul(so3_exp(d[0], d[1], d[2]), load_q(knots + 4 * i)));
  }
  return CMX_OK;
}

int cmx_traj_evaluate(int order, int K, const double *knots, int64_t start_ns, int64_t dt_ns, int64_t t_ns,
                      double quat_out[4]) {
  if (!knots || !quat_out || dt_ns <= 0 || K < order) return CMX_ERR_INVALID_ARG;
  if (order == 2) return evaluate_quat<2>(K, knots, start_ns, dt_ns, t_ns, quat_out);
  if (order == 4) return evaluate_quat<4>(K, knots, start_ns, dt_ns, t_ns, quat_out);
  return CMX_ERR_INVALID_ARG;
}

int cmx_bearing_lut(int W, int H, const double K[9], const double D[5], const double R[9], const double P[12], double *lut) {
  if (W <= 0 || H <= 0 || !K || !lut) return CMX_ERR_INVALID_ARG;
  static const double kEye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  static const double kNoDist[5] = {0, 0, 0, 0, 0};
  if (!D) D = kNoDist;
  if (!R) R = kEye;
  const double Pk[12] = {K[0], K[1], K[2], 0, K[3], K[4], K[5], 0, K[6], K[7], K[8], 0};
  if (!P) P = Pk;
  if (K[0] == 0 || K[4] == 0 || P[0] == 0 || P[5] == 0) return CMX_ERR_INVALID_ARG;
  const bool distorted = D[0] != 0 || D[1] != 0 || D[2] != 0 || D[3] != 0 || D[4] != 0;
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  for (int row = 0; row < H; row++)
    for (int col = 0; col < W; col++) {
      double u = col, v = row;  // rectifyPoint: identity without distortion
      if (distorted) {
        // the raw pixel goes through cv::undistortPoints as an fp32 point: five fixed-point sweeps of the
        // plumb_bob model in fp64, then R and P, stored back as fp32
        double x = ((double)(float)col - K[2]) / K[0], y = ((double)(float)row - K[5]) / K[4];
        const double xd = x, yd = y;
        for (int sweep = 0; sweep < 5; sweep++) {
          const double r2 = x * x + y * y;
          const double inv_radial = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
          const double tx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
          const double ty = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
          x = (xd - tx) * inv_radial;
          y = (yd - ty) * inv_radial;
        }
        const double X = R[0] * x + R[1] * y + R[2], Y = R[3] * x + R[4] * y + R[5], Z = R[6] * x + R[7] * y + R[8];
        u = (double)(float)((X / Z) * P[0] + P[2]);
        v = (double)(float)((Y / Z) * P[5] + P[6]);
      }
      double *o = lut + 3 * ((size_t)row * W + col);  // projectPixelTo3dRay on the projection matrix
      o[0] = (u - P[2] - P[3]) / P[0];
      o[1] = (v - P[6] - P[7]) / P[5];
      o[2] = 1.0;
    }
  return CMX_OK;
}

}  // extern "C"
