/*
 * traj_init.c -- control-pose initialisation (front-end angular velocities -> back-end control poses) and the
 * bearing-vector look-up table, restated.  TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).
 *
 * Follows  src/backend/pose_graph_optimizer.cpp:191-222      integrateAngVel
 *          src/backend/trajectory.cpp:112-192 (linear), :357-464 (cubic)   fitCtrlPoses
 *          src/backend/trajectory.cpp:205-214, :480-489       generateCtrlPoses (number of control poses)
 *          src/cmax_slam.cpp:106-120                          precomputeBearingVectors
 *          Eigen (vendored: thirdparty/basalt-headers/thirdparty/eigen)
 *              Eigen/src/QR/FullPivHouseholderQR.h:457-537 (computeInPlace), :542-573 (_solve_impl), :246-255 (rank)
 *              Eigen/src/Householder/Householder.h:65-94 (makeHouseholder), :113-130 (applyHouseholderOnTheLeft)
 * PINNED: orc_fullpiv_qr_solve against the vendored Eigen itself (oracle/_ref/libbasalt_ref.so, ref_fullpiv_qr_solve)
 *         and the SO(3) exp/log/product against the vendored Sophus (same library).
 * PARITY UNPINNED: the bearing LUT -- image_geometry::PinholeCameraModel::{rectifyPoint, projectPixelTo3dRay} and
 *         cv::undistortPoints are external, un-vendored code; restated from their documented behaviour (noetic /
 *         OpenCV 4.2: fp32 point round trip, 5 fixed-point iterations, plumb_bob k1 k2 p1 p2 k3).
 */
#include "cmax_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ros::Duration::toSec() of an exact ns difference (normalizeSecNSecSigned: nsec in [0,1e9)) */
static double dur_to_sec(int64_t d) {
  int64_t s = d / 1000000000LL, n = d % 1000000000LL;
  if (n < 0) { n += 1000000000LL; s -= 1; }
  return (double)s + 1e-9 * (double)n;
}

/* integrateAngVel  pose_graph_optimizer.cpp:191-222 */
int orc_integrate_ang_vel(int n, const int64_t *t_ns, const double *ang_vel, int64_t pose_t_ns,
                          const double pose_quat[4], int64_t *prev_t_ns, double prev_ang_vel[3],
                          int first_time_window, int64_t *out_t_ns, double *out_quat) {
  int64_t cur_t = pose_t_ns;
  double cur_q[4] = {pose_quat[0], pose_quat[1], pose_quat[2], pose_quat[3]};
  int m = 0;
  for (int i = 0; i < n; i++) {
    /* :199-203  skip data whose stamp is not newer than the previous one */
    if (!(t_ns[i] > *prev_t_ns) && !first_time_window) continue;
    /* :205-206 */
    const double dt = dur_to_sec(t_ns[i] - cur_t);
    double drotv[3];
    for (int k = 0; k < 3; k++) drotv[k] = dt * ((prev_ang_vel[k] + ang_vel[3 * i + k]) / 2.0);
    /* :209-210  post-multiplication */
    double e[4], r[4];
    orc_so3_exp(drotv, e);
    orc_so3_mul(cur_q, e, r);
    memcpy(cur_q, r, sizeof r);
    cur_t = t_ns[i];
    /* :213  std::map insert: an existing key is kept (cannot happen for strictly increasing stamps) */
    if (m > 0 && out_t_ns[m - 1] == cur_t) {
      /* keep the first */
    } else {
      out_t_ns[m] = cur_t;
      memcpy(out_quat + 4 * m, cur_q, sizeof cur_q);
      m++;
    }
    /* :216 */
    *prev_t_ns = t_ns[i];
    for (int k = 0; k < 3; k++) prev_ang_vel[k] = ang_vel[3 * i + k];
  }
  return m;
}

/* generateCtrlPoses: std::round((t_end - t_beg).toSec()/dt_knots_) + 1 (linear) / + 3 (cubic) */
int orc_num_ctrl_poses(int order, int64_t t_beg_ns, int64_t t_end_ns, double dt_knots) {
  return (int)round(dur_to_sec(t_end_ns - t_beg_ns) / dt_knots) + (order == 4 ? 3 : 1);
}

/* ------------------------------------------------------------------------------------------------------------
 * x = A.fullPivHouseholderQr().solve(b),  A rows x cols column-major-agnostic (we take row-major input).
 * Returns the rank Eigen would report. */
int orc_fullpiv_qr_solve(int rows, int cols, const double *A_rowmajor, const double *b, double *x) {
  const int size = rows < cols ? rows : cols;
  double *qr = (double *)malloc(sizeof(double) * rows * cols); /* qr[i*cols+j] */
  double *h = (double *)calloc(size > 0 ? size : 1, sizeof(double));
  int *rt = (int *)malloc(sizeof(int) * (size > 0 ? size : 1)), *ct = (int *)malloc(sizeof(int) * (size > 0 ? size : 1));
  double *c = (double *)malloc(sizeof(double) * rows);
  int *perm = (int *)malloc(sizeof(int) * cols);
  memcpy(qr, A_rowmajor, sizeof(double) * rows * cols);
  memcpy(c, b, sizeof(double) * rows);
#define Q(i, j) qr[(size_t)(i) * cols + (j)]
  const double eps = 2.220446049250313e-16;
  const double precision = eps * (double)size; /* m_precision :471 */
  double biggest = 0, maxpivot = 0;
  int nonzero_pivots = size;
  for (int k = 0; k < size; k++) {
    /* :488-493  biggest |entry| of the bottom-right corner; Eigen's maxCoeff visitor runs column by column
     * (column-major storage) and keeps the first maximum */
    int rb = k, cb = k;
    double best = -1.0;
    for (int j = k; j < cols; j++)
      for (int i = k; i < rows; i++) {
        const double a = fabs(Q(i, j));
        if (a > best) { best = a; rb = i; cb = j; }
      }
    if (k == 0) biggest = best;
    /* :497  isMuchSmallerThan(x, y, prec): |x| <= |y| * prec */
    if (best <= biggest * precision) {
      nonzero_pivots = k;
      for (int i = k; i < size; i++) { rt[i] = i; ct[i] = i; h[i] = 0; }
      break;
    }
    rt[k] = rb; ct[k] = cb;
    if (k != rb) /* :512  only the tail(cols-k) of the two rows is swapped */
      for (int j = k; j < cols; j++) { double t = Q(k, j); Q(k, j) = Q(rb, j); Q(rb, j) = t; }
    if (k != cb)
      for (int i = 0; i < rows; i++) { double t = Q(i, k); Q(i, k) = Q(i, cb); Q(i, cb) = t; }
    /* makeHouseholderInPlace on column k, rows k..  (Householder.h:65-94) */
    double tail_sq = 0;
    for (int i = k + 1; i < rows; i++) tail_sq += Q(i, k) * Q(i, k);
    const double c0 = Q(k, k);
    double tau, beta;
    if (tail_sq <= 2.2250738585072014e-308) {
      tau = 0; beta = c0;
      for (int i = k + 1; i < rows; i++) Q(i, k) = 0;
    } else {
      beta = sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < rows; i++) Q(i, k) = Q(i, k) / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    h[k] = tau;
    Q(k, k) = beta;
    if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
    /* :527  apply H = I - tau v v^T (v = [1, essential]) to the remaining columns
     * Householder.h:113-130: tmp = essential^T * bottom; tmp += row0; row0 -= tau*tmp; bottom -= tau*essential*tmp */
    for (int j = k + 1; j < cols; j++) {
      if (rows - k == 1) { Q(k, j) *= 1.0 - tau; continue; }
      if (tau == 0) continue;
      double tmp = 0;
      for (int i = k + 1; i < rows; i++) tmp += Q(i, k) * Q(i, j);
      tmp += Q(k, j);
      Q(k, j) -= tau * tmp;
      for (int i = k + 1; i < rows; i++) Q(i, j) -= tau * Q(i, k) * tmp;
    }
  }
  /* :531-533  column permutation from the transpositions */
  for (int j = 0; j < cols; j++) perm[j] = j;
  for (int k = 0; k < size; k++) { int t = perm[k]; perm[k] = perm[ct[k]]; perm[ct[k]] = t; }
  /* rank()  :246-255 */
  int rank = 0;
  {
    const double thr = fabs(maxpivot) * (eps * (double)size);
    for (int i = 0; i < nonzero_pivots; i++) rank += fabs(Q(i, i)) > thr;
  }
  /* _solve_impl  :542-573 */
  if (rank == 0) {
    for (int j = 0; j < cols; j++) x[j] = 0;
  } else {
    for (int k = 0; k < rank; k++) {
      { double t = c[k]; c[k] = c[rt[k]]; c[rt[k]] = t; }
      if (rows - k == 1) { /* Householder.h:118  a one-row block is scaled by (1 - tau) */
        c[k] *= 1.0 - h[k];
      } else if (h[k] != 0) {
        double tmp = 0;
        for (int i = k + 1; i < rows; i++) tmp += Q(i, k) * c[i];
        tmp += c[k];
        c[k] -= h[k] * tmp;
        for (int i = k + 1; i < rows; i++) c[i] -= h[k] * Q(i, k) * tmp;
      }
    }
    for (int i = rank - 1; i >= 0; i--) { /* upper-triangular back substitution */
      double s = c[i];
      for (int j = i + 1; j < rank; j++) s -= Q(i, j) * c[j];
      c[i] = s / Q(i, i);
    }
    for (int i = 0; i < rank; i++) x[perm[i]] = c[i];
    for (int i = rank; i < cols; i++) x[perm[i]] = 0;
  }
#undef Q
  free(qr); free(h); free(rt); free(ct); free(c); free(perm);
  return rank;
}

/* fitCtrlPoses  trajectory.cpp:112-192 (order 2) / :357-464 (order 4).  Returns 0, or -1 where the reference's
 * CHECK_GE / Eigen index assertions would fire. */
int orc_fit_ctrl_poses(int order, int n_poses, const int64_t *t_ns, const double *quat, double t_beg, double dt_knots,
                       int num_cps, double *out_quat) {
  if (n_poses < num_cps || num_cps < order || (order != 2 && order != 4)) return -1; /* CHECK_GE :116 / :387 */
  static const double M2[4] = {1.0, 0.0, -1.0, 1.0};
  static const double M4[16] = {1. / 6, 2. / 3, 1. / 6, 0.0, -0.5, 0.0, 0.5, 0.0,
                                0.5,    -1.0,   0.5,    0.0, -1. / 6, 0.5, -0.5, 1. / 6};
  const double *M = order == 2 ? M2 : M4;
  /* 1. lift: increments w.r.t. the first pose */
  const double off[4] = {quat[0], quat[1], quat[2], quat[3]};
  const double off_inv[4] = {-off[0], -off[1], -off[2], off[3]};
  double *N = (double *)calloc((size_t)n_poses * num_cps, sizeof(double));
  double *D = (double *)malloc(sizeof(double) * 3 * n_poses); /* Dx | Dy | Dz */
  int rc = 0;
  for (int p = 0; p < n_poses; p++) {
    double dq[4], w[3];
    orc_so3_mul(off_inv, quat + 4 * p, dq);
    const double t = orc_time_to_sec(t_ns[p]);
    const int t_i = (int)floor((t - t_beg) / dt_knots);
    const double u = (t - (t_i * dt_knots + t_beg)) / dt_knots;
    if (t_i < 0 || t_i + order > num_cps) { rc = -1; break; } /* Eigen index assertion */
    double U[4];
    for (int i = 0; i < order; i++) U[i] = pow(u, i);
    for (int j = 0; j < order; j++) {
      double s = 0; /* Eigen 1xn * nxn product: sum over i in order */
      for (int i = 0; i < order; i++) s += U[i] * M[i * order + j];
      N[(size_t)p * num_cps + t_i + j] = s;
    }
    orc_so3_log(dq, w);
    D[p] = w[0]; D[n_poses + p] = w[1]; D[2 * n_poses + p] = w[2];
  }
  if (rc == 0) {
    double *P = (double *)malloc(sizeof(double) * 3 * num_cps);
    for (int a = 0; a < 3; a++) orc_fullpiv_qr_solve(n_poses, num_cps, N, D + (size_t)a * n_poses, P + (size_t)a * num_cps);
    /* 3. retract */
    for (int i = 0; i < num_cps; i++) {
      const double drotv[3] = {P[i], P[num_cps + i], P[2 * num_cps + i]};
      double e[4];
      orc_so3_exp(drotv, e);
      orc_so3_mul(off, e, out_quat + 4 * i);
    }
    free(P);
  }
  free(N); free(D);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * precomputeBearingVectors  cmax_slam.cpp:106-120:
 *   rectified = cam.rectifyPoint(Point2d(x,y));  bearing = cam.projectPixelTo3dRay(rectified)
 * image_geometry (noetic) restated from its documented behaviour -- PARITY UNPINNED:
 *   rectifyPoint: if D is all zero (distortion_state NONE) return the raw point; otherwise convert the point to
 *     fp32, cv::undistortPoints(src, dst, K, D, R, P), return the fp32 result widened to double.
 *   cv::undistortPoints (OpenCV 4.2, default criteria = 5 iterations): x=(u-cx)/fx, y=(v-cy)/fy; x0=x,y0=y;
 *     5x { r2=x^2+y^2; icdist=1/(1+((k3 r2+k2) r2+k1) r2); dx=2 p1 x y+p2 (r2+2x^2); dy=p1 (r2+2y^2)+2 p2 x y;
 *          x=(x0-dx) icdist; y=(y0-dy) icdist };  [X Y W]=R [x y 1];  x=X/W, y=Y/W;  u'=x P00+P02, v'=y P11+P12
 *     computed in fp64 and stored to the fp32 destination.
 *   projectPixelTo3dRay: ((u-cx'-Tx)/fx', (v-cy'-Ty)/fy', 1) with fx'=P00, fy'=P11, cx'=P02, cy'=P12,
 *     Tx=P03, Ty=P13.
 * K 3x3, R 3x3, P 3x4 row-major; D = k1 k2 p1 p2 k3. */
void orc_bearing_lut(int W, int H, const double K[9], const double D[5], const double R[9], const double P[12],
                     double *lut) {
  const int distorted = D[0] != 0 || D[1] != 0 || D[2] != 0 || D[3] != 0 || D[4] != 0;
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  for (int yy = 0; yy < H; yy++)
    for (int xx = 0; xx < W; xx++) {
      double ur = (double)xx, vr = (double)yy;
      if (distorted) {
        const float u32 = (float)xx, v32 = (float)yy;
        double x = ((double)u32 - cx) / fx, y = ((double)v32 - cy) / fy;
        const double x0 = x, y0 = y;
        for (int it = 0; it < 5; it++) {
          const double r2 = x * x + y * y;
          const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
          const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
          const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
          x = (x0 - dx) * icdist;
          y = (y0 - dy) * icdist;
        }
        const double X = R[0] * x + R[1] * y + R[2], Y = R[3] * x + R[4] * y + R[5], Wv = R[6] * x + R[7] * y + R[8];
        const double xn = X / Wv, yn = Y / Wv;
        ur = (double)(float)(xn * P[0] + P[2]);
        vr = (double)(float)(yn * P[5] + P[6]);
      }
      double *o = lut + 3 * ((size_t)yy * W + xx);
      o[0] = (ur - P[2] - P[3]) / P[0];
      o[1] = (vr - P[6] - P[7]) / P[5];
      o[2] = 1.0;
    }
}
