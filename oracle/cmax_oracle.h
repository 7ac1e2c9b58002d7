/*
 * cmax_oracle.h -- CPU restatement of cmax_slam's event-warping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the "port" CPU
 * baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product path (cmax_slam_amd/, libcmaxhip.so) never
 * links, imports or calls anything in oracle/.
 *
 * Plain C11, single-threaded, fp64 geometry / fp32 images exactly as the
 * reference.  Every function cites the reference file:line it restates
 * (paths relative to the reference checkout).
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - SO(3) spline value + Jacobians: PINNED against the reference's own
 *     vendored Basalt/Sophus code compiled from /root/reference
 *     (oracle/_ref/libbasalt_ref.so, recipe in oracle/Makefile) and against
 *     committed golden vectors produced by it (tests/golden/spline_*.npz).
 *   - IWE / blur / contrast / gradient: the reference ships no tests or
 *     fixtures for these and its translation units need ROS + OpenCV (absent):
 *     PARITY UNPINNED at the OpenCV/ROS boundary; OpenCV semantics
 *     (GaussianBlur, meanStdDev, mean, MatExpr) are restated from OpenCV 4.2's
 *     documented behaviour.
 */
#ifndef CMAX_ORACLE_H
#define CMAX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* contrast measures: include/frontend/local_focus_funcs.h:7-11,
 * include/backend/global_focus_funcs.h */
enum { ORC_VARIANCE = 0, ORC_MEAN_SQUARE = 1, ORC_GRADIENT_MAGNITUDE = 2 };

/* ---- OpenCV semantics restated (cv_ops.c) ---- */
int  orc_gauss_ksize(double sigma);                       /* cv::GaussianBlur Size(0,0) rule, CV_32F */
void orc_gauss_kernel(int ksize, double sigma, float *k); /* cv::getGaussianKernel(ksize, sigma, CV_32F) */
/* in-place separable blur of an interleaved cn-channel fp32 image, BORDER_REFLECT_101 */
void orc_gaussian_blur(float *img, int W, int H, int cn, double sigma);
/* contrast + gradient over P derivative planes; plane k element i is ch[k][i*stride]. */
double orc_contrast(const float *img, int npix, const float *const *ch, int stride, int P,
                    int measure, int W, int H, double *grad /* P or NULL */);

/* ---- ros::Time arithmetic restated (time_ops.c) ---- */
/* time_batch = time_first + (time_last-time_first)*0.5  (ros::Duration::operator*(double) -> fromSec) */
int64_t orc_time_batch_ns(int64_t t_first_ns, int64_t t_last_ns);
double  orc_time_to_sec(int64_t t_ns);                    /* ros::Time::toSec() */

/* ---- front end (frontend.c) ---- */
typedef struct {
  int W, H;            /* sensor size */
  const double *lut;   /* W*H*3 bearing vectors, index (y*W+x)*3   (cmax_slam.cpp:106-120) */
  double fx, fy, cx, cy;
  int batch;           /* event_batch_size */
  double sigma;        /* blur_sigma */
  int measure;         /* contrast_measure */
} orc_fe_cfg;

/* AngVelEstimator::computeImageOfWarpedEvents (local_image_warped_events.cpp:10-39 / :41-57).
 * iwe: H*W fp32; deriv: H*W*3 interleaved fp32 or NULL; blur!=0 applies the Gaussian (first overload). */
int orc_fe_iwe(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y,
               const int64_t *t_ns, int64_t t_ref_ns, const double omega[3],
               float *iwe, float *deriv, int blur);
/* one batch of warpAndAccumulateEvents (local_image_warped_events.cpp:59-170); events [beg, end) */
void orc_fe_warp_batch(const orc_fe_cfg *c, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                       int64_t beg, int64_t end, int64_t t_ref_ns, const double omega[3], float *iwe, float *deriv);
/* local_contrast_fdf body without the sign flip (local_optim_contrast_gsl.cpp:20-56):
 * returns contrast, grad[3] if non-NULL. */
int orc_fe_eval(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y,
                const int64_t *t_ns, int64_t t_ref_ns, const double omega[3],
                double *contrast, double *grad);

/* ---- SO(3) spline (so3_spline.c) ---- */
/* basalt::So3Spline<order>::evaluate (so3_spline.h:218-274). knots: K quaternions (x,y,z,w).
 * R: row-major 3x3. J: order blocks of row-major 3x3 (d_val_d_knot[i]) or NULL.
 * returns 0, or -1 if the Basalt asserts would fire. */
int orc_so3_spline_eval(int order, int K, const double *knots_xyzw, int64_t start_ns, int64_t dt_ns,
                        int64_t t_ns, double *quat_xyzw_out, double *R, double *J, int *start_idx);
/* knot_i <- exp(drot) * knot_i  (trajectory.cpp:221-238 / :491-499) */
void orc_so3_left_update(double *knot_xyzw, const double drot[3]);
void orc_so3_exp(const double w[3], double q_xyzw[4]);
void orc_so3_log(const double q_xyzw[4], double w[3]);
void orc_so3_mul(const double a_xyzw[4], const double b_xyzw[4], double out_xyzw[4]);
/* (double)->ns truncation of CopyAndIncrementalUpdate (trajectory.cpp:255-256, :58-67) */
int64_t orc_traj_temp_start_ns(double t_beg, int idx_traj_beg, double dt_knots);

/* ---- back end (backend.c) ---- */
typedef struct {
  int W, H;            /* sensor */
  const double *lut;
  int Wp, Hp;          /* panorama */
  int batch, sample_rate;
  double sigma;
  int measure;
  int order;           /* 2 linear / 4 cubic */
  int K;               /* knots in the temp trajectory */
  int64_t start_ns, dt_ns;
  int num_fixed;       /* num_cps_fixed_ */
  int64_t t_next_win_beg_ns;
} orc_be_cfg;

typedef struct {
  /* persistent EventWarper state across evaluations of one window
   * (event_pano_warper.h:82-90,104,114) */
  float *IL_old, *IL_new, *IL, *IG, *IGp; /* Hp*Wp each, caller-allocated */
  double alpha;
  int first_iter;
} orc_be_state;

/* EventWarper::computeImageOfWarpedEvents (event_pano_warper.cpp:167-231) with the temp trajectory
 * already updated (knots = K quaternions after CopyAndIncrementalUpdate).
 * iwe: Hp*Wp; planes: P=3*(K-num_fixed) planes of Hp*Wp contiguous, or NULL. */
int orc_be_iwe(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
               const int64_t *t_ns, const double *knots_xyzw, float *iwe, float *planes);
/* one batch of EventWarper::warpAndAccumulateEvents (event_pano_warper.cpp:233-336); votes into st->IL_old / IL_new */
int orc_be_warp_batch(const orc_be_cfg *c, orc_be_state *st, const uint16_t *x, const uint16_t *y,
                      const int64_t *t_ns, int64_t beg, int64_t end, const double *knots_xyzw, float *planes);
/* global_contrast_fdf body without sign flip (global_optim_contrast_gsl_analytical.cpp:17-68):
 * knots0 = temp-trajectory knots before the update; drotv: 3*(K-num_fixed). */
int orc_be_eval(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
                const int64_t *t_ns, const double *knots0_xyzw, const double *drotv,
                double *contrast, double *grad, float *iwe_out /* optional */);
/* EventWarper::updateAlpha (event_pano_warper.cpp:134-165) */
double orc_be_alpha(const float *IGp, const float *IL, int npix);
/* global-map upkeep, once per window (event_pano_warper.cpp:81-126) */
void orc_be_update_ig(float *IG, const float *IL_old, const uint8_t *update_times, int npix, int max_update_times);
void orc_be_mark_visited(int W, int H, const double *lut, int Wp, int Hp, const double quat_xyzw[4], int radius,
                         uint8_t *update_times);
/* dvs::EquirectangularCamera::projectToImage (equirectangular_camera.h:18-45) */
void orc_equirect_project(int Wp, int Hp, const double P[3], double px[2], float jac[6] /* or NULL */);

/* ---- control-pose initialisation + bearing LUT (traj_init.c) ---- */
/* PoseGraphOptimizer::integrateAngVel (pose_graph_optimizer.cpp:191-222): n stamped angular velocities (sorted),
 * the latest pose, the previous angular velocity (in/out).  Returns the number of poses written. */
int orc_integrate_ang_vel(int n, const int64_t *t_ns, const double *ang_vel, int64_t pose_t_ns,
                          const double pose_quat[4], int64_t *prev_t_ns, double prev_ang_vel[3],
                          int first_time_window, int64_t *out_t_ns, double *out_quat);
/* generateCtrlPoses' count (trajectory.cpp:205-214 / :480-489) */
int orc_num_ctrl_poses(int order, int64_t t_beg_ns, int64_t t_end_ns, double dt_knots);
/* Linear/CubicTrajectory::fitCtrlPoses (trajectory.cpp:112-192 / :357-464) */
int orc_fit_ctrl_poses(int order, int n_poses, const int64_t *t_ns, const double *quat, double t_beg, double dt_knots,
                       int num_cps, double *out_quat);
/* Eigen's A.fullPivHouseholderQr().solve(b) restated; returns the rank */
int orc_fullpiv_qr_solve(int rows, int cols, const double *A_rowmajor, const double *b, double *x);
/* CMaxSLAM::precomputeBearingVectors (cmax_slam.cpp:106-120) over image_geometry + cv::undistortPoints */
void orc_bearing_lut(int W, int H, const double K[9], const double D[5], const double R[9], const double P[12],
                     double *lut);

/* ---- all-cores OpenMP variant (allcores.c, liboracle_mt.so only) -- NOT the reference, which is single-threaded:
 * contiguous batch ranges per thread, thread-private images summed in thread order.  nthreads <= 0: all cores. */
int orc_mt_max_threads(void);
int orc_fe_eval_mt(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                   int64_t t_ref_ns, const double omega[3], int nthreads, double *contrast, double *grad);
int orc_be_eval_mt(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
                   const int64_t *t_ns, const double *knots0_xyzw, const double *drotv, int nthreads,
                   double *contrast, double *grad);

#ifdef __cplusplus
}
#endif
#endif
