/*
 * frontend.c -- front-end Image-of-Warped-Events + contrast, restated line by line.
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).
 *
 * Follows  src/frontend/local_image_warped_events.cpp:10-170
 *          src/utils/image_geom_util.cpp:7-41, include/utils/image_geom_util.h:5-8
 *          src/frontend/local_focus_funcs.cpp:82-120
 *          src/frontend/local_optim_contrast_gsl.cpp:20-56
 * fp64 geometry, fp32 weights and accumulators, sequential event order, exactly as the reference.
 * Compile with -ffp-contract=off (the reference is built for baseline x86-64: no FMA).
 */
#include "cmax_oracle.h"
#include <stdlib.h>
#include <string.h>

/* cv::Matx<double,m,n> = a(m x l) * b(l x n):  s = 0; s += a(i,k)*b(k,j)  (Matx_MatMulOp) */
static void matx_mul_d(const double *a, const double *b, double *out, int m, int l, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < l; k++) s += a[i * l + k] * b[k * n + j];
      out[i * n + j] = s;
    }
}

/* warpAndAccumulateEvents  local_image_warped_events.cpp:59-170 */
void orc_fe_warp_batch(const orc_fe_cfg *c, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                       int64_t beg, int64_t end, int64_t t_ref_ns, const double omega[3], float *iwe,
                       float *deriv) {
  /* :68-76  shared pose for the batch */
  const int64_t time_batch = orc_time_batch_ns(t_ns[beg], t_ns[end - 1]);
  const double dt = orc_time_to_sec(time_batch) - orc_time_to_sec(t_ref_ns);
  const double drot[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt}; /* ang_vel * dt */
  const int W = c->W, H = c->H;

  for (int64_t e = beg; e < end; e++) {
    /* :100-101  p' = p + delta_rot x p */
    const double *p = c->lut + 3 * ((size_t)y[e] * W + x[e]);
    const double px = p[0], py = p[1], pz = p[2];
    const double rx = px + (drot[1] * pz - drot[2] * py);
    const double ry = py + (drot[2] * px - drot[0] * pz);
    const double rz = pz + (drot[0] * py - drot[1] * px);

    /* :121-122  canonicalProjection  image_geom_util.cpp:24-41 */
    const double inverse_depth = 1.0 / rz;
    const double cxn = rx * inverse_depth, cyn = ry * inverse_depth;

    /* :130  applyIntrinsics  image_geom_util.cpp:7-22 */
    const double u = c->fx * cxn + c->cx;
    const double v = c->fy * cyn + c->cy;

    double J[6]; /* jacobian_warped_pt_wrt_ang_vel, 2x3 */
    if (deriv) {
      /* :110  cross2Matrix((-dt)*point_3D)  image_geom_util.h:5-8 */
      const double vx = (-dt) * px, vy = (-dt) * py, vz = (-dt) * pz;
      const double Jrot[9] = {0, -vz, vy, vz, 0, -vx, -vy, vx, 0};
      const double Jproj[6] = {inverse_depth, 0.0, -cxn * inverse_depth, 0.0, inverse_depth, -cyn * inverse_depth};
      double Jcal[6];
      matx_mul_d(Jproj, Jrot, Jcal, 2, 3, 3); /* :126 */
      const double Jpix[4] = {c->fx, 0., 0., c->fy};
      matx_mul_d(Jpix, Jcal, J, 2, 2, 3); /* :135 */
    }

    /* :139-145 */
    const int xx = (int)u, yy = (int)v;
    if (1 <= xx && xx < W - 2 && 1 <= yy && yy < H - 2) {
      const float dx = (float)(u - xx), dy = (float)(v - yy);
      float *r0p = iwe + (size_t)yy * W + xx, *r1p = r0p + W;
      r0p[0] += (1.f - dx) * (1.f - dy);
      r0p[1] += dx * (1.f - dy);
      r1p[0] += (1.f - dx) * dy;
      r1p[1] += dx * dy;
      if (deriv) {
        /* :157-166  cv::Point3f r0, r1; Point3f*float + Point3f*float accumulated in fp32 */
        const float r0[3] = {(float)J[0], (float)J[1], (float)J[2]};
        const float r1[3] = {(float)J[3], (float)J[4], (float)J[5]};
        float *d00 = deriv + 3 * ((size_t)yy * W + xx), *d01 = d00 + 3;
        float *d10 = d00 + 3 * (size_t)W, *d11 = d10 + 3;
        for (int k = 0; k < 3; k++) {
          d00[k] += r0[k] * (-(1.f - dy)) + r1[k] * (-(1.f - dx));
          d01[k] += r0[k] * (1.f - dy) + r1[k] * (-dx);
          d10[k] += r0[k] * (-dy) + r1[k] * (1.f - dx);
          d11[k] += r0[k] * dy + r1[k] * dx;
        }
      }
    }
  }
}

/* computeImageOfWarpedEvents  local_image_warped_events.cpp:10-39 (blur=1) / :41-57 (blur=0, deriv NULL) */
int orc_fe_iwe(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
               int64_t t_ref_ns, const double omega[3], float *iwe, float *deriv, int blur) {
  const size_t np = (size_t)c->W * c->H;
  for (int64_t i = 0; i < n; i++)
    if (x[i] >= c->W || y[i] >= c->H) return -1; /* the reference's .at() would throw */
  memset(iwe, 0, np * sizeof(float));
  if (deriv) memset(deriv, 0, np * 3 * sizeof(float));
  for (int64_t beg = 0; beg < n; beg += c->batch) { /* :22-28 */
    int64_t end = beg + c->batch;
    if (end > n) end = n;
    orc_fe_warp_batch(c, x, y, t_ns, beg, end, t_ref_ns, omega, iwe, deriv);
  }
  if (blur && c->sigma > 0) { /* :32-38 */
    orc_gaussian_blur(iwe, c->W, c->H, 1, c->sigma);
    if (deriv) orc_gaussian_blur(deriv, c->W, c->H, 3, c->sigma);
  }
  return 0;
}

/* local_contrast_fdf  local_optim_contrast_gsl.cpp:20-56 (caller flips the sign) */
int orc_fe_eval(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                int64_t t_ref_ns, const double omega[3], double *contrast, double *grad) {
  const size_t np = (size_t)c->W * c->H;
  float *iwe = (float *)malloc(np * sizeof(float));
  float *deriv = grad ? (float *)malloc(np * 3 * sizeof(float)) : NULL;
  int rc = orc_fe_iwe(c, n, x, y, t_ns, t_ref_ns, omega, iwe, deriv, 1);
  if (rc == 0) {
    const float *ch[3] = {deriv, deriv ? deriv + 1 : NULL, deriv ? deriv + 2 : NULL}; /* cv::split :93 */
    *contrast = orc_contrast(iwe, (int)np, ch, 3, 3, c->measure, c->W, c->H, grad);
  }
  free(iwe);
  free(deriv);
  return rc;
}
