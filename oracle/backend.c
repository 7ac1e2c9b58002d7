/*
 * backend.c -- back-end panoramic Image-of-Warped-Events + contrast, restated line by line.
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).
 *
 * Follows  src/backend/event_pano_warper.cpp:128-336
 *          include/backend/equirectangular_camera.h:11-45,64-67
 *          src/backend/trajectory.cpp:86-110,221-263,329-355,491-522
 *          src/backend/global_focus_funcs.cpp:52-80
 *          src/backend/global_optim_contrast_gsl_analytical.cpp:17-68
 * fp64 rotation/projection, fp32 Jacobian chain / weights / accumulators, sequential event order.
 * Compile with -ffp-contract=off.
 */
#include "cmax_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CV_PI 3.1415926535897932384626433832795

/* dvs::EquirectangularCamera::projectToImage  equirectangular_camera.h:18-45; focalFromFOV :64-67
 * (hfov=360, vfov=180: event_pano_warper.cpp:31) */
void orc_equirect_project(int Wp, int Hp, const double P[3], double px[2], float jac[6]) {
  const double fx = (double)((Wp / 360.0) * 180.0 / CV_PI);
  const double fy = (double)((Hp / 180.0) * 180.0 / CV_PI);
  const double cxp = (double)Wp / 2.0, cyp = (double)Hp / 2.0;
  const double x = P[0], y = P[1], z = P[2];
  const double phi = atan2(x, z);
  const double theta = asin(y / sqrt(x * x + y * y + z * z));
  const double rho = sqrt(x * x + y * y + z * z); /* P.norm() */
  const double Ydivrho = y / rho;
  if (jac) {
    const double XdivZ = x / z;
    const double tmp1 = fx / ((1 + XdivZ * XdivZ) * z);
    const double tmp2 = -fy / sqrt(1 - Ydivrho * Ydivrho);
    const double tmp3 = Ydivrho / (rho * rho);
    jac[0] = (float)tmp1;
    jac[1] = 0;
    jac[2] = (float)(-tmp1 * XdivZ);
    jac[3] = (float)(tmp2 * tmp3 * x);
    jac[4] = (float)(tmp2 * (tmp3 * y - 1 / rho));
    jac[5] = (float)(tmp2 * tmp3 * z);
  }
  px[0] = cxp + phi * fx;
  px[1] = cyp + theta * fy;
}

/* EventWarper::updateAlpha  event_pano_warper.cpp:134-165 (lambda0 = 1) */
double orc_be_alpha(const float *IGp, const float *IL, int npix) {
  int nz = 0;
  for (int i = 0; i < npix; i++) nz += (IGp[i] != 0.f);
  if (nz < 1) return 0; /* :137-141 */
  double area_g = 0, num_g = 0, area_l = 0, num_l = 0;
  for (int i = 0; i < npix; i++) {
    const float eg = expf(-1.0f * IGp[i]); /* cv::exp on the fp32 image -(1/lambda0)*IGp */
    area_g += (double)(1.f - eg);          /* cv::sum: fp64 accumulation */
    num_g += (double)IGp[i];
    const float el = expf(-1.0f * IL[i]);
    area_l += (double)(1.f - el);
    num_l += (double)IL[i];
  }
  const double dens_g = num_g / area_g, dens_l = num_l / area_l;
  return dens_l / dens_g;
}

/* EventWarper::warpAndAccumulateEvents  event_pano_warper.cpp:233-336 */
int orc_be_warp_batch(const orc_be_cfg *c, orc_be_state *st, const uint16_t *x, const uint16_t *y,
                      const int64_t *t_ns, int64_t beg, int64_t end, const double *knots, float *planes) {
  const int n3 = 3 * c->order;
  const size_t np = (size_t)c->Wp * c->Hp;
  /* :239-242 */
  const int64_t time_batch = orc_time_batch_ns(t_ns[beg], t_ns[end - 1]);
  /* :250-256  traj->evaluate -> So3Spline::evaluate; Jacobian blocks copied into a 3 x 3n fp32 cv::Mat
   * with block k at columns 3k..3k+2 (trajectory.cpp:99-106 / :342-351) */
  double R[9], Jd[9 * 6];
  int idx_cp_beg = 0;
  if (orc_so3_spline_eval(c->order, c->K, knots, c->start_ns, c->dt_ns, time_batch, NULL, R, planes ? Jd : NULL,
                          &idx_cp_beg) != 0)
    return -2;
  float Jcp[3 * 12]; /* ddrot_ddrot_cp: 3 x 3n, row-major */
  if (planes)
    for (int k = 0; k < c->order; k++)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Jcp[j * n3 + 3 * k + i] = (float)Jd[9 * k + j * 3 + i];

  for (int64_t e = beg; e < end; e += c->sample_rate) { /* :262 */
    const double *b = c->lut + 3 * ((size_t)y[e] * c->W + x[e]);
    /* :269  e_ray_w = R * e_ray_cam */
    double ew[3];
    for (int i = 0; i < 3; i++) ew[i] = R[3 * i] * b[0] + R[3 * i + 1] * b[1] + R[3 * i + 2] * b[2];
    double px[2];
    float dpm_drb[6];
    orc_equirect_project(c->Wp, c->Hp, ew, px, planes ? dpm_drb : NULL); /* :274 */

    float jac[2 * 12];
    if (planes) {
      /* :280-285 */
      double rb[3];
      for (int i = 0; i < 3; i++) { /* cv::Matx33d * Point3d: s = 0; s += a(i,k)*b(k) */
        double s = 0;
        for (int k = 0; k < 3; k++) s += R[3 * i + k] * b[k];
        rb[i] = s;
      }
      const float drb[9] = {0, (float)rb[2], (float)-rb[1], (float)-rb[2], 0, (float)rb[0], (float)rb[1], (float)-rb[0], 0};
      float dpm_ddrot[6]; /* Matx23f * Matx33f, fp32 accumulation */
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
          float s = 0;
          for (int k = 0; k < 3; k++) s += dpm_drb[i * 3 + k] * drb[k * 3 + j];
          dpm_ddrot[i * 3 + j] = s;
        }
      /* Matx23f * cv::Mat(3 x 3n, CV_32F) -> cv::gemm: fp64 accumulation, fp32 result */
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < n3; j++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += (double)dpm_ddrot[i * 3 + k] * (double)Jcp[k * n3 + j];
          jac[i * n3 + j] = (float)s;
        }
    }

    /* :290-296 */
    const int xx = (int)px[0], yy = (int)px[1];
    const float dx = (float)(px[0] - xx), dy = (float)(px[1] - yy);
    if (1 <= xx && xx < c->Wp - 2 && 1 <= yy && yy < c->Hp - 2) {
      float *img = (t_ns[e] < c->t_next_win_beg_ns) ? st->IL_old : st->IL_new; /* :298-311 */
      float *r0p = img + (size_t)yy * c->Wp + xx, *r1p = r0p + c->Wp;
      r0p[0] += (1.f - dx) * (1.f - dy);
      r0p[1] += dx * (1.f - dy);
      r1p[0] += (1.f - dx) * dy;
      r1p[1] += dx * dy;
      if (planes) {
        for (int i = 0; i < n3; i++) { /* :316-332 */
          const float r0 = jac[i], r1 = jac[n3 + i];
          const int j = 3 * (idx_cp_beg - c->num_fixed) + i;
          if (j >= 0) {
            float *d0 = planes + (size_t)j * np + (size_t)yy * c->Wp + xx, *d1 = d0 + c->Wp;
            d0[0] += r0 * (-(1.f - dy)) + r1 * (-(1.f - dx));
            d0[1] += r0 * (1.f - dy) + r1 * (-dx);
            d1[0] += r0 * (-dy) + r1 * (1.f - dx);
            d1[1] += r0 * dy + r1 * dx;
          }
        }
      }
    }
  }
  return 0;
}

/* EventWarper::computeImageOfWarpedEvents  event_pano_warper.cpp:167-231 */
int orc_be_iwe(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
               const int64_t *t_ns, const double *knots, float *iwe, float *planes) {
  const size_t np = (size_t)c->Wp * c->Hp;
  const int P = 3 * (c->K - c->num_fixed);
  for (int64_t i = 0; i < n; i++)
    if (x[i] >= c->W || y[i] >= c->H) return -1;
  memset(st->IL_old, 0, np * sizeof(float)); /* :173-174 */
  memset(st->IL_new, 0, np * sizeof(float));
  if (planes) memset(planes, 0, np * (size_t)P * sizeof(float)); /* :176-185 */

  /* :188-196  for (beg = begin; beg < end-1; beg += B)  -- a trailing single-event batch is skipped */
  for (int64_t beg = 0; beg < n - 1; beg += c->batch) {
    const int64_t left = n - beg;
    const int64_t end = (left > c->batch) ? beg + c->batch : n;
    int rc = orc_be_warp_batch(c, st, x, y, t_ns, beg, end, knots, planes);
    if (rc) return rc;
  }
  for (size_t i = 0; i < np; i++) st->IL[i] = st->IL_old[i] + st->IL_new[i]; /* :199 cv::add */
  if (st->first_iter) { /* :201-210 */
    memcpy(st->IGp, st->IG, np * sizeof(float)); /* updateIGp :128-132 */
    st->alpha = orc_be_alpha(st->IGp, st->IL, (int)np);
    st->first_iter = 0;
  }
  { /* :213 cv::scaleAdd(IGp, alpha, IL, iwe): fp32  IGp*(float)alpha + IL */
    const float a = (float)st->alpha;
    for (size_t i = 0; i < np; i++) iwe[i] = st->IGp[i] * a + st->IL[i];
  }
  if (c->sigma > 0) { /* :217-230 */
    orc_gaussian_blur(iwe, c->Wp, c->Hp, 1, c->sigma);
    if (planes)
      for (int k = 0; k < P; k++) orc_gaussian_blur(planes + (size_t)k * np, c->Wp, c->Hp, 1, c->sigma);
  }
  return 0;
}

/* global_contrast_fdf  global_optim_contrast_gsl_analytical.cpp:17-68 (caller flips the sign) */
int orc_be_eval(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
                const int64_t *t_ns, const double *knots0, const double *drotv, double *contrast, double *grad,
                float *iwe_out) {
  const size_t np = (size_t)c->Wp * c->Hp;
  const int Kopt = c->K - c->num_fixed;
  const int P = 3 * Kopt;
  /* copyAndUpdateTraj -> CopyAndIncrementalUpdate -> incrementalUpdate (trajectory.cpp:240-263 / :501-522) */
  double *knots = (double *)malloc(sizeof(double) * 4 * (size_t)c->K);
  memcpy(knots, knots0, sizeof(double) * 4 * (size_t)c->K);
  for (int i = c->num_fixed; i < c->K; i++) orc_so3_left_update(knots + 4 * i, drotv + 3 * (i - c->num_fixed));

  float *iwe = iwe_out ? iwe_out : (float *)malloc(np * sizeof(float));
  float *planes = grad ? (float *)malloc(np * (size_t)P * sizeof(float)) : NULL;
  int rc = orc_be_iwe(c, st, n, x, y, t_ns, knots, iwe, planes);
  if (rc == 0) {
    const float **ch = (const float **)malloc(sizeof(float *) * (size_t)(P > 0 ? P : 1));
    for (int k = 0; k < P; k++) ch[k] = planes ? planes + (size_t)k * np : NULL;
    *contrast = orc_contrast(iwe, (int)np, ch, 1, P, c->measure == ORC_MEAN_SQUARE ? ORC_MEAN_SQUARE : ORC_VARIANCE,
                             c->Wp, c->Hp, grad);
    free(ch);
  }
  if (!iwe_out) free(iwe);
  free(planes);
  free(knots);
  return rc;
}

/* ---- global-map upkeep (once per window, outside the optimiser loop) ---- */

/* EventWarper::updateIG  event_pano_warper.cpp:109-126:
 * IG(y,x) += IL_old(y,x) wherever the visit count is <= max_update_times */
void orc_be_update_ig(float *IG, const float *IL_old, const uint8_t *update_times, int npix, int max_update_times) {
  for (int i = 0; i < npix; i++)
    if ((int)update_times[i] <= max_update_times) IG[i] += IL_old[i];
}

/* EventWarper::setUpdateTimesIG  event_pano_warper.cpp:81-107 (+ warpEventToMap :37-54):
 * every sensor pixel is warped with `rot` onto the panorama; a (2*radius+1)^2 neighbourhood of the hit cell is
 * marked in a mask; the mask is added (cv::add on CV_8U: saturating) to the visit-count map.
 * The reference's row test is `0 <= y_mask + j` (sic) and it then indexes mask.at(y_mask, x_mask) -- for
 * y_mask < 0 that is an out-of-bounds write; this restatement keeps the published test AND skips y_mask < 0. */
void orc_be_mark_visited(int W, int H, const double *lut, int Wp, int Hp, const double quat_xyzw[4], int radius,
                         uint8_t *update_times) {
  const double qx = quat_xyzw[0], qy = quat_xyzw[1], qz = quat_xyzw[2], qw = quat_xyzw[3];
  /* rot.matrix(): Eigen toRotationMatrix */
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                       txz - twy, tyz + twx, 1 - (txx + tyy)};
  uint8_t *mask = (uint8_t *)calloc((size_t)Wp * Hp, 1);
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const double *b = lut + 3 * ((size_t)y * W + x);
      double e[3], px[2];
      for (int i = 0; i < 3; i++) e[i] = R[3 * i] * b[0] + R[3 * i + 1] * b[1] + R[3 * i + 2] * b[2];
      orc_equirect_project(Wp, Hp, e, px, NULL);
      const int ic = (int)px[0], ir = (int)px[1];
      for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
          const int x_mask = ic + i, y_mask = ir + j;
          if (0 <= y_mask + j && y_mask < Hp && 0 <= x_mask && x_mask < Wp && y_mask >= 0)
            mask[(size_t)y_mask * Wp + x_mask] = 1;
        }
    }
  for (size_t i = 0; i < (size_t)Wp * Hp; i++) {
    const int v = (int)update_times[i] + (int)mask[i];
    update_times[i] = (uint8_t)(v > 255 ? 255 : v);
  }
  free(mask);
}
