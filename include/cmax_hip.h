/*
 * cmax_hip.h -- C ABI of libcmaxhip.so: MI355X (gfx950) evaluator for cmax_slam's event-warping hot path.
 *
 * This is the drop-in boundary.  The reference sits behind GSL's
 *   gsl_multimin_function_fdf { f, df, fdf, n, params }
 * filled at src/frontend/local_optim_contrast_gsl.cpp:87-96 and src/backend/global_optim_contrast_gsl.cpp:23-33.
 * The bodies of local_contrast_{f,df,fdf} / global_contrast_{f,df,fdf} become ~10-line calls into the entry
 * points below (INTEGRATION.md shows them); everything they used to compute on the CPU
 *   (computeImageOfWarpedEvents + computeContrast) runs as hand-written HIP kernels.
 *
 * Conventions
 *   - plain C types only; every entry point returns an int status (CMX_OK == 0) and never aborts the host
 *     (the reference's glog CHECK / Basalt assert / std::out_of_range abort paths become error codes).
 *   - a context owns all device memory; inputs are copied at set_*; outputs go to caller buffers.
 *   - one context per path (front end / back end); contexts are independent and may be driven from two
 *     host threads concurrently (src/node.cpp:22 + src/cmax_slam.cpp:92); one context is not thread-safe.
 *   - eval() is synchronous: it returns when contrast / gradient are on the host.
 *   - contrast and gradient are returned with the reference's sign (maximised quantity); the GSL glue
 *     negates them exactly as local_optim_contrast_gsl.cpp:48-55 does.
 *   - quaternions are (x, y, z, w) (Eigen coeffs() order); matrices row-major.
 */
#ifndef CMAX_HIP_H
#define CMAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: exactly these symbols are exported */
#endif

typedef struct cmx_ctx cmx_ctx;

enum {
  CMX_OK = 0,
  CMX_ERR_INVALID_ARG = 1,  /* null pointer, bad size, unsupported spline order, ... */
  CMX_ERR_EVENT_RANGE = 2,  /* an event has x >= W or y >= H (reference: std::out_of_range from .at()) */
  CMX_ERR_HIP = 3,          /* a HIP runtime call failed; see cmx_last_error() */
  CMX_ERR_SPLINE_RANGE = 4, /* a batch time lies outside the knot support (reference: BASALT_ASSERT abort) */
  CMX_ERR_STATE = 5,        /* eval before set_packet / set_window, wrong context kind, ... */
  CMX_ERR_TIME_ORDER = 6    /* a batch spans a negative time interval (reference: CHECK_GE abort) */
};

/* contrast_measure: include/frontend/local_focus_funcs.h:7-11.  As in the reference's switch statements any other
 * value means VARIANCE, and the back end (global_focus_funcs.cpp:61-69) treats GRADIENT_MAGNITUDE as VARIANCE too.
 * GRADIENT_MAGNITUDE (Sobel, front end) always uses the derivative-plane gradient. */
enum { CMX_VARIANCE = 0, CMX_MEAN_SQUARE = 1, CMX_GRADIENT_MAGNITUDE = 2 };

/* how the analytic gradient is formed */
enum {
  CMX_GRAD_PLANES = 0, /* faithful: scatter P derivative planes, blur them, reduce (reference data flow) */
  CMX_GRAD_ADJOINT = 1 /* equivalent: blur is linear => grad_k = sum_events <dW_k, G^T 2(G I - mu)/N>; one extra
                          pass over the events gathers from one plane; no derivative planes */
};

/* cmx_set_option keys a HOST may need: CMX_OPT_DETERMINISTIC (bitwise reproducibility), CMX_OPT_SPIN_WAIT (how its threads wait) and
 * CMX_OPT_GRAD_MODE / CMX_OPT_SPLAT_MODE together to select the reference-shaped data flow.  That is the supported surface.  The A/B
 * switches that tests and same-box measurements use to reach every internal form on any input (forms the library otherwise selects BY
 * ITSELF from the configuration, defaulting to the fastest) live in include/cmax_hip_diag.h: same cmx_set_option, no stability promise. */
enum {
  CMX_OPT_GRAD_MODE = 1,  /* CMX_GRAD_ADJOINT (default) | CMX_GRAD_PLANES */
  CMX_OPT_SPLAT_MODE = 2, /* 0 = one global fp32 atomic per vote;
                             1 (default) = LDS-privatised: events are sorted once per packet/window by the 32x32 destination
                                 tile of their vote, workgroups accumulate in LDS and flush touched pixels; votes that
                                 leave a window (parameters drifted) take the global path, so results stay exact;
                                 applies to the plane-0 splat (cost-only evaluations and CMX_GRAD_ADJOINT) */
  CMX_OPT_DETERMINISTIC = 5, /* 1: bitwise run-to-run reproducible results (default 0).  With the LDS-privatised splat
                             (CMX_OPT_SPLAT_MODE 1, adjoint gradient or cost-only) every vote that reaches global memory
                             becomes a 64-bit integer add into a 2^-30 fixed-point plane -- integer adds commute -- and
                             one extra pass converts it to the fp32 plane; the front-end gather then walks the events in
                             time order and large panoramas do not use the compacted tile list, so that every
                             floating-point sum has a fixed order.  Costs ~10-20 % per evaluation.  The reference-shaped
                             flow (derivative planes, fp32 atomics) stays order-dependent */
  CMX_OPT_SPIN_WAIT = 4   /* ONE policy for the three places a host thread of this library waits:
                             (a) an evaluation waiting for its last kernel -- spins on a completion ticket that kernel writes to
                                 mapped host memory after the results (a few microseconds sooner than hipStreamSynchronize returns);
                             (b) a group's worker threads between two commands;  (c) a CMX_SCHED_BACKGROUND context held behind an
                                 urgent burst (cmx_set_sched_class).
                             1 (default): (a) spins for as long as the evaluation runs (the calling thread is blocked in a
                                 synchronous call either way: one core busy for its ~40-250 us); (b), (c) -- threads with NOTHING
                                 on the device -- spin for 50 us, then sleep on a condition variable: an idle group and a held back
                                 end use no core.
                             0: never spin: (a) is plain hipStreamSynchronize, (b) / (c) sleep at once.
                             n >= 2: spin budget in microseconds for all three ((a): then hipStreamSynchronize). */
};

const char *cmx_version(void);
int cmx_device_count(void);
const char *cmx_last_error(const cmx_ctx *ctx);
const char *cmx_status_string(int status);
void cmx_destroy(cmx_ctx *ctx);
int cmx_set_option(cmx_ctx *ctx, int key, int value);
/* run the context's work on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int cmx_set_stream(cmx_ctx *ctx, void *hip_stream);
/* Two paths on one GPU.  The reference runs the front end (a packet every 10 ms, src/node.cpp:22) beside the back-end thread's
 * window solves (src/cmax_slam.cpp:92); on one GPU the two contexts' kernels share the compute units.  (Stream priorities and CU masks
 * were measured not to deliver the asked-for pair of slow-downs, profiles/r04_fe_beside_be.txt: the two entry points live in
 * cmax_hip_diag.h since ABI 6.) */
/* What DOES give the front end its latency back without a static split: cooperative scheduling between the contexts of one
 * process that share a device, at evaluation granularity.  A context of class CMX_SCHED_BACKGROUND (the back end) holds its
 * NEXT evaluation -- between two evaluations of a solve there is nothing of it on the GPU -- while a context of class
 * CMX_SCHED_URGENT on the same device (the front end: a 0.5 ms solve every 10 ms) has a call of cmx_*_eval / _eval_each /
 * _solve in progress, or finished one less than 20 us ago (so that a GSL-driven sequence of evaluations counts as one burst).
 * The urgent call then waits for at most the one background evaluation already on the device.  A background context never
 * waits longer than 5 ms in a row.  Host-side only: no kernel, stream or result changes.  Default: CMX_SCHED_NORMAL (neither
 * waits nor is waited for).  Measured: bench.py frontend_beside_backend.cooperative. */
enum { CMX_SCHED_BACKGROUND = -1, CMX_SCHED_NORMAL = 0, CMX_SCHED_URGENT = 1 };
int cmx_set_sched_class(cmx_ctx *ctx, int sched_class);

/* ------------------------------------------------------------------ front end -------------------------
 * replaces AngVelEstimator::computeImageOfWarpedEvents + computeContrast
 *   (src/frontend/local_image_warped_events.cpp:10-170, src/frontend/local_focus_funcs.cpp:82-120)
 * as called from local_contrast_fdf (src/frontend/local_optim_contrast_gsl.cpp:20-56). */

/* lut: W*H*3 fp64 bearing vectors, index (y*W+x)*3 -- the reference's precomputed_bearing_vectors
 * (src/cmax_slam.cpp:106-120); copied to the device once. */
int cmx_frontend_create(cmx_ctx **out, int device, int W, int H, const double *lut);

/* ------------------------------------------------------------------ events as the host holds them (AoS) ------
 * The reference keeps its events as std::vector<dvs_msgs::Event> (src/frontend/ang_vel_estimator.cpp:68-147: pushEvent / the
 * per-packet copy into event_subset_; src/backend/pose_graph_optimizer.cpp:131-165: getEventSubset) -- an array of
 *   struct { uint16_t x, y; struct { uint32_t sec, nsec; } ts; uint8_t polarity; }        (16 bytes)
 * The *_aos entry points take that memory as it is: ONE packing pass on the host pool straight from the array into the pinned
 * upload buffer (x | y << 16 [| old << 31]; t_ns = sec * 1e9 + nsec formed on the fly for the batch times / the store's
 * timestamps), no intermediate x[] / y[] / t_ns[] vectors.  `layout` describes any array of records: record size and the byte
 * offsets of the four fields (uint16 x, uint16 y, uint32 sec, uint32 nsec); CMX_AOS_DVS_EVENT is dvs_msgs::Event's.  Same
 * checks, same errors, bit-identical device contents as the SoA entry points given the same events. */
typedef struct { size_t stride, off_x, off_y, off_sec, off_nsec; } cmx_aos_layout;
#define CMX_AOS_DVS_EVENT {16, 0, 2, 4, 8}

/* State AngVelEstimator hands over before a solve (src/frontend/ang_vel_estimator.cpp:137-147):
 * event_subset_ (SoA here; polarity is never read), time_packet_, camera_matrix_ (fx,fy,cx,cy),
 * warp_opt.{event_batch_size, blur_sigma}, process_opt.contrast_measure.  Uploads once; every
 * evaluation of the solve reuses it. */
int cmx_frontend_set_packet(cmx_ctx *ctx, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                            int64_t t_ref_ns, double fx, double fy, double cx, double cy, int event_batch_size,
                            double blur_sigma, int contrast_measure);
int cmx_frontend_set_packet_aos(cmx_ctx *ctx, int64_t n, const void *events, const cmx_aos_layout *layout, int64_t t_ref_ns,
                                double fx, double fy, double cx, double cy, int event_batch_size, double blur_sigma,
                                int contrast_measure);

/* Packet pipeline (no reference counterpart: the CPU path has no set-up to hide).  Queues everything the packet's first
 * evaluation would otherwise do before its first vote -- destination-tile sort at omega_hint, tile-ordered bearing / dt
 * streams, chunk table -- behind the upload of cmx_frontend_set_packet[_from], and returns without waiting.  A host with
 * two contexts calls set_packet + prepare for packet k+1 on one of them BEFORE it solves packet k on the other; the GPU
 * then runs upload and sort beside the solve.  omega_hint: any estimate of the solve's
 * starting point (the previous packet's result); it affects speed only, never results. */
int cmx_frontend_prepare(cmx_ctx *ctx, const double omega_hint[3]);

/* local_contrast_fdf body: contrast (and d contrast / d omega if grad != NULL; grad == NULL is the cost-only
 * fast path used by local_contrast_f, src/frontend/local_optim_contrast_gsl.cpp:58-63). */
int cmx_frontend_eval(cmx_ctx *ctx, const double omega[3], double *contrast, double *grad /* [3] or NULL */);

/* Line-search hint (used by cmx_frontend_solve / cmx_backend_solve; available to any host whose line search knows its
 * acceptance test in advance).  Call right before a COST-ONLY evaluation: `mode` says under which condition on that
 * evaluation's value f = -contrast the gradient at the same point will be requested next -- 1: f < threshold (GSL
 * conjugate_fr's trial step: fc < fa), 2: f <= threshold (Brent loop: fm <= fb), 3: !(f >= threshold) (bracketing loop),
 * 4: always, 0: withdraw the hint.  The evaluator then queues the gradient pass behind the cost evaluation, gated on the
 * device by the cost it has just computed; a following cmx_*_eval(same parameters, grad != NULL) finds its result in flight
 * and saves one host-to-GPU turnaround (~6 us).  The hint holds for one evaluation; results never depend on it.
 * Adjoint gradient with CMX_OPT_REUSE_IMAGE and CMX_OPT_TAIL_FINALIZE on, no communicator; ignored otherwise. */
int cmx_hint_next_df(cmx_ctx *ctx, double threshold, int mode);

/* m INDEPENDENT evaluations in one call: omegas = m x 3, contrasts = m, grads = m x 3 or NULL (cost-only).  The m launch
 * chains are queued back to back and the host waits once, so the per-evaluation host round trip (~3 us of ~42) and the
 * GPU idle time behind it disappear; results are those of m cmx_frontend_eval calls.  For candidate lists (multi-start,
 * grid initialisation, finite differences) -- the line search of local_optim_contrast_gsl.cpp cannot use it, each of its
 * trial points depends on the previous cost.  Not available with a communicator attached. */
int cmx_frontend_eval_many(cmx_ctx *ctx, int m, const double *omegas, double *contrasts, double *grads);

/* exactly m calls of cmx_frontend_eval, one after the other (each waited for before the next is issued), without returning
 * to the caller in between: the evaluation pattern of the optimiser loop (local_optim_contrast_gsl.cpp:125-176) for hosts
 * whose own per-call overhead is large (an interpreter) -- replaying a recorded sequence, timing the evaluator itself. */
int cmx_frontend_eval_each(cmx_ctx *ctx, int m, const double *omegas, double *contrasts, double *grads);

/* computeImageOfWarpedEvents for display / inspection: iwe = H*W fp32 (required), deriv = H*W*3 interleaved
 * fp32 (CV_32FC3 layout) or NULL.  blur = 0 is the display overload (local_image_warped_events.cpp:41-57). */
int cmx_frontend_get_iwe(cmx_ctx *ctx, const double omega[3], int blur, float *iwe, float *deriv);

/* ------------------------------------------------------------------ back end --------------------------
 * replaces PoseGraphOptimizer::copyAndUpdateTraj + EventWarper::computeImageOfWarpedEvents + computeContrast
 *   (src/backend/trajectory.cpp:240-263,501-522; src/backend/event_pano_warper.cpp:128-336;
 *    src/backend/global_focus_funcs.cpp:52-80)
 * as called from global_contrast_fdf (src/backend/global_optim_contrast_gsl_analytical.cpp:17-68). */

/* Wp x Hp panorama: fx = Wp/2pi, fy = Hp/pi (include/backend/equirectangular_camera.h:64-67). */
int cmx_backend_create(cmx_ctx **out, int device, int W, int H, const double *lut, int Wp, int Hp);

/* State PoseGraphOptimizer::processTimeWindow hands over (src/backend/pose_graph_optimizer.cpp:283-293):
 *   window events; the temp trajectory CopyAndIncrementalUpdate builds = knots [idx_cp_traj_beg_, size) of traj_
 *   with start_ns = int64(1e9*(t_beg_ + idx_cp_traj_beg_*dt_knots_)) (cmx_traj_temp_start_ns) and dt_ns;
 *   order = 2 (linear, So3Spline<2>) or 4 (cubic, So3Spline<4>); num_fixed = idx_cp_opt_beg_ - idx_cp_traj_beg_;
 *   t_next_win_beg = t_win_beg_ + win_stride_; warp_opt_.{event_batch_size, event_sample_rate, blur_sigma};
 *   IG = the persistent global map (Hp*Wp fp32) or NULL for an all-zero map.  Also performs setFirstIter(true):
 *   alpha is recomputed by the first evaluation of the window (event_pano_warper.cpp:201-210). */
int cmx_backend_set_window(cmx_ctx *ctx, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                           int order, int K, const double *knots_xyzw, int64_t start_ns, int64_t dt_ns,
                           int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate,
                           double blur_sigma, int contrast_measure, const float *IG);
int cmx_backend_set_window_aos(cmx_ctx *ctx, int64_t n, const void *events, const cmx_aos_layout *layout, int order, int K,
                               const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                               int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG);

/* Window pipeline: the back end's counterpart of cmx_frontend_prepare -- pose table at drotv_hint (NULL = zero increments),
 * destination-tile sort, chunk table and bearing streams of the window handed over last, queued without waiting. */
int cmx_backend_prepare(cmx_ctx *ctx, const double *drotv_hint /* [3*(K-num_fixed)] or NULL */);

/* global_contrast_fdf body: drotv = 3*(K-num_fixed) incremental rotation vectors applied by LEFT
 * multiplication to the non-fixed knots (trajectory.cpp:236 / :497); grad has the same length or is NULL. */
int cmx_backend_eval(cmx_ctx *ctx, const double *drotv, double *contrast, double *grad);

/* m independent evaluations in one call (see cmx_frontend_eval_many): drotvs = m x 3(K-num_fixed), grads likewise or NULL */
int cmx_backend_eval_many(cmx_ctx *ctx, int m, const double *drotvs, double *contrasts, double *grads);
/* m calls of cmx_backend_eval, one after the other (see cmx_frontend_eval_each) */
int cmx_backend_eval_each(cmx_ctx *ctx, int m, const double *drotvs, double *contrasts, double *grads);

enum { CMX_PLANE_IL_OLD = 0, CMX_PLANE_IL_NEW = 1, CMX_PLANE_IWE = 2, CMX_PLANE_DERIV0 = 16 };
/* planes of the LAST evaluation (what updateIG / publishEventImage read, event_pano_warper.cpp:109-126):
 * IL_old / IL_new are raw; IWE is the blurred I = IL + alpha*IGp; CMX_PLANE_DERIV0+j is blurred derivative
 * plane j (only after an evaluation with CMX_GRAD_PLANES and grad != NULL). host: Hp*Wp fp32. */
int cmx_backend_get_plane(cmx_ctx *ctx, int which, float *host);
int cmx_backend_get_alpha(cmx_ctx *ctx, double *alpha);
/* Parity read-back of the per-batch pose table the splat and gather kernels read (be_pose_table kernels = So3Spline::evaluate
 * with Jacobians, thirdparty/basalt-headers/include/basalt/spline/so3_spline.h:218-274, as Trajectory::evaluate stores them,
 * src/backend/trajectory.cpp:86-110 / :329-355): recomputed on the device at the LAST evaluation's parameters (zero increments
 * before the first one).  R = n x 9 fp64 row-major; Jcp = n x 36 fp32, the 3 x 3*order row-major block in front; idx = first
 * control pose of the batch's segment; t_batch_ns = the batch's pose time.  Any output may be NULL; *n_batches = batches of
 * the window; at most max_batches rows are written. */
int cmx_backend_get_pose_table(cmx_ctx *ctx, int max_batches, double *R, float *Jcp, int *idx, int64_t *t_batch_ns, int *n_batches);

/* Global-map upkeep on the device (once per window, after the solve).  Keeps IG_ and
 * IG_update_times_map_ resident across windows: pass IG = CMX_KEEP_MAP to cmx_backend_set_window instead of
 * round-tripping a 4..32 MB plane through the host every window.
 *   cmx_backend_update_map    EventWarper::updateIG          (src/backend/event_pano_warper.cpp:109-126): IG += IL_old of
 *                             the LAST evaluation wherever the visit count is <= max_update_times
 *   cmx_backend_mark_visited  EventWarper::setUpdateTimesIG  (:81-107) for one pose (the caller loops over the poses
 *                             every 0.05 s, src/backend/pose_graph_optimizer.cpp:325-337); counts saturate at 255
 *   cmx_backend_reset_map     resetIG + zero visit counts;   get/set_map: host copies (either pointer may be NULL) */
#define CMX_KEEP_MAP ((const float *)(uintptr_t)1)
int cmx_backend_update_map(cmx_ctx *ctx, int max_update_times);
int cmx_backend_mark_visited(cmx_ctx *ctx, const double quat_xyzw[4], int radius);
int cmx_backend_reset_map(cmx_ctx *ctx);
int cmx_backend_get_map(cmx_ctx *ctx, float *IG, unsigned char *visits);
int cmx_backend_set_map(cmx_ctx *ctx, const float *IG, const unsigned char *visits);
/* int64_t(1e9 * (t_beg + idx_traj_beg*dt_knots)) -- the (double)->ns truncation of trajectory.cpp:255-256 */
int64_t cmx_traj_temp_start_ns(double t_beg, int idx_traj_beg, double dt_knots);

/* ------------------------------------------------------------------ device-resident event store -----------
 * SURVEY.md section 8f rank 3.  The reference keeps the stream in AngVelEstimator::events_ and copies a packet
 * (src/frontend/ang_vel_estimator.cpp:137-147) or a window (src/backend/pose_graph_optimizer.cpp:131-165) out of it for
 * every solve; consecutive packets and windows overlap heavily.  With a store the stream is uploaded ONCE
 * (cmx_events_push as events arrive), packets / windows are cut from it on the device by global event index, and
 * cmx_events_drop_before is deleteOldEvents (ang_vel_estimator.cpp:149-173).  Results are identical to
 * cmx_frontend_set_packet / cmx_backend_set_window on the same events.  One store per GPU, shared by the front-end and
 * back-end contexts of that GPU; not thread-safe (serialise push/drop against set_*_from, as the reference does with
 * mutex_events).
 * cmx_events_create_group: ONE store with a replica of the stream on every distinct device of a group's member list (the list
 * given to cmx_backend_create_group): cmx_events_push packs once on the host and uploads to all replicas side by side,
 * cmx_events_drop_before compacts all of them, cmx_backend_set_window_from on the group's handle lets every member cut ITS
 * batch range on its own device (no event crosses the host at hand-over), and any front-end / back-end context on one of
 * those devices can cut from it too.  cmx_events_devices lists the replicas' devices (returns their number). */
typedef struct cmx_events cmx_events;
int cmx_events_create(cmx_events **out, int device, int W, int H, size_t capacity);
int cmx_events_create_group(cmx_events **out, const int *devices, int n_devices, int W, int H, size_t capacity);
int cmx_events_devices(const cmx_events *ev, int *devices, int max_devices);
void cmx_events_destroy(cmx_events *ev);
const char *cmx_events_last_error(const cmx_events *ev);
int cmx_events_push(cmx_events *ev, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns);
int cmx_events_push_aos(cmx_events *ev, int64_t n, const void *events, const cmx_aos_layout *layout); /* e.g. msg->events.data() */
int cmx_events_drop_before(cmx_events *ev, int64_t global_index);
int64_t cmx_events_begin(const cmx_events *ev); /* global index of the oldest event held */
int64_t cmx_events_end(const cmx_events *ev);   /* one past the newest */
int cmx_frontend_set_packet_from(cmx_ctx *ctx, const cmx_events *ev, int64_t first, int64_t count, int64_t t_ref_ns,
                                 double fx, double fy, double cx, double cy, int event_batch_size, double blur_sigma,
                                 int contrast_measure);
int cmx_backend_set_window_from(cmx_ctx *ctx, const cmx_events *ev, int64_t first, int64_t count, int order, int K,
                                const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed,
                                int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate,
                                double blur_sigma, int contrast_measure, const float *IG);

/* ------------------------------------------------------------------ split-phase (multi-GPU) ------------
 * The IWE is a sum over events, the contrast a non-linear function of the SUMMED image, so ranks exchange
 * between splat and blur/reduce.  Each rank loads its contiguous range of event
 * batches with set_packet / set_window, then per evaluation:
 *     cmx_*_accumulate(ctx, x, want_grad)     zero + splat the rank's partial planes
 *     all-reduce(sum) cmx_accum_ptr(ctx) [cmx_accum_count floats]      (RCCL, caller-side)
 *     cmx_*_finish(ctx, &contrast, grad)      blur + reduce on the summed planes
 *   with CMX_GRAD_ADJOINT the planes are I only and finish() returns the rank's PARTIAL gradient
 *   (sum over its own events): all-reduce(sum) that small vector too.
 * cmx_set_accum_buffer lets the caller own the accumulation memory (e.g. a torch tensor, so
 * torch.distributed can all-reduce it in place); it must hold cmx_accum_capacity(ctx) floats. */
size_t cmx_accum_capacity(const cmx_ctx *ctx);             /* floats needed for the worst case of this context */
int cmx_set_accum_buffer(cmx_ctx *ctx, void *device_ptr, size_t n_floats);
void *cmx_accum_ptr(const cmx_ctx *ctx);                   /* device pointer of the accumulation planes */
size_t cmx_accum_count(const cmx_ctx *ctx);                /* floats written by the last accumulate() */
int cmx_frontend_accumulate(cmx_ctx *ctx, const double omega[3], int want_grad);
int cmx_frontend_finish(cmx_ctx *ctx, double *contrast, double *grad);
int cmx_backend_accumulate(cmx_ctx *ctx, const double *drotv, int want_grad);
int cmx_backend_finish(cmx_ctx *ctx, double *contrast, double *grad);
/* With CMX_GRAD_ADJOINT the gradient needs a second, tiny exchange (each rank gathers over its own events from the
 * common Itilde plane).  finish() = finish_begin() + finish_end(); between the two the rank's partial gradient sums
 * sit in device memory at cmx_grad_ptr(ctx) [cmx_grad_count(ctx) doubles] for an in-place all-reduce(sum); nothing
 * is synchronised with the host until finish_end().  cmx_set_grad_buffer makes that buffer caller-owned. */
int cmx_frontend_finish_begin(cmx_ctx *ctx, int want_grad);
int cmx_frontend_finish_end(cmx_ctx *ctx, double *contrast, double *grad);
int cmx_backend_finish_begin(cmx_ctx *ctx, int want_grad);
int cmx_backend_finish_end(cmx_ctx *ctx, double *contrast, double *grad);
void *cmx_grad_ptr(const cmx_ctx *ctx);
size_t cmx_grad_count(const cmx_ctx *ctx);
int cmx_set_grad_buffer(cmx_ctx *ctx, void *device_ptr, size_t n_doubles);

/* Native exchange: attach an RCCL communicator to the context (one process per GPU) and every cmx_*_eval / cmx_*_solve
 * performs the all-reduces itself, in place, on the context's stream -- partial planes after the splat, and the 2P
 * partial gradient sums after the gather pass (adjoint mode).  All ranks therefore see identical contrast / gradient
 * and take identical optimiser decisions.  Every rank must use the same options (cmx_set_option) and issue the same
 * sequence of calls; which collectives an evaluation issues then depends on rank-invariant state only (plane size,
 * options, call sequence) -- never on how many events a rank happens to hold (an empty shard takes part like any other).
 * Back-end planes of 1 MB and more are exchanged as a SET OF TILES (64 x 16 pixels): the tiles exchanged -- packed from both
 * planes into one staging buffer with the tile-occupancy map behind them, one collective -- are those any rank flagged in
 * the PREVIOUS evaluation, dilated by one tile in every direction (the whole planes on a window's first evaluation, or when
 * the set exceeds half of the map) -- no host synchronisation between splat and blur.  A kernel lists the flagged tiles the set
 * did not cover; if there are any (parameters jumped) the evaluation is completed by exchanging exactly those and finishing again.
 * Smaller planes travel whole.  cmx_get_stats reports host synchronisations inside sharded evaluations (0), misses and set size.
 * Rank 0 creates the 128-byte id (cmx_comm_unique_id); the launcher distributes it by whatever means it has
 * (torch.distributed broadcast in bench.py, MPI, a file).  RCCL is dlopen()ed at this point only; hosts that never
 * attach a communicator do not need it installed. */
#define CMX_COMM_ID_BYTES 128
int cmx_comm_unique_id(char id[CMX_COMM_ID_BYTES]);
int cmx_comm_attach(cmx_ctx *ctx, const char id[CMX_COMM_ID_BYTES], int rank, int nranks);
int cmx_comm_detach(cmx_ctx *ctx);
/* The same exchange points over a caller-supplied transport (MPI, a shared-memory ring, a test harness): `fn` must
 * all-reduce `count` elements at `device_buf` IN PLACE across the nranks participants, ordered after the work already
 * queued on `hip_stream` and before work queued afterwards (i.e. enqueue on that stream, or synchronise it, exchange, and
 * return), and return 0 on success.  Ranks call it in the same order with the same count / dtype / op. */
enum { CMX_DT_U8 = 0, CMX_DT_F32 = 1, CMX_DT_F64 = 2 };
enum { CMX_OP_SUM = 0, CMX_OP_MAX = 1 };
typedef int (*cmx_allreduce_fn)(void *user, void *device_buf, size_t count, int dtype, int op, void *hip_stream);
int cmx_comm_attach_custom(cmx_ctx *ctx, cmx_allreduce_fn fn, void *user, int rank, int nranks);
/* What is attached, AS THE COMMUNICATOR ITSELF REPORTS IT: *transport = 0 none, 1 RCCL (rank / nranks from ncclCommUserRank /
 * ncclCommCount of the live communicator -- not the numbers passed at attach), 2 caller-supplied or a group's direct transport
 * (the numbers given at attach).  On a group handle: member 0's communicator.  Any pointer may be NULL. */
int cmx_comm_info(cmx_ctx *ctx, int *rank, int *nranks, int *transport);

/* ------------------------------------------------------------------ one-process multi-GPU: a GROUP ---------
 * The reference's host is ONE process with ONE back-end thread and ONE GSL instance (src/cmax_slam.cpp:92,
 * src/backend/global_optim_contrast_gsl.cpp:23-33): it cannot be started once per GPU.  cmx_backend_create_group returns an
 * ordinary back-end handle whose evaluations fan out to n_devices member contexts: cmx_backend_set_window shards the window
 * by whole event batches (member r = rank r of the one-process-per-GPU form; event_pano_warper.cpp:188-196 is the loop being
 * split), cmx_backend_eval / cmx_backend_solve run the members' splat, the plane / tile-set exchange, blur and gather on all
 * devices, add the members' gradients on the calling thread (the gradient is linear in the members' row sums: no second
 * collective) and return ONE contrast / gradient -- the bodies of global_contrast_{f,df,fdf} do not
 * change, there is one optimiser and no launcher.  The map upkeep calls, cmx_set_option and cmx_destroy act on every member
 * (each keeps its own replica of IG); cmx_backend_get_plane / get_alpha / get_map / cmx_get_stats read member 0.  The caller
 * stays single-threaded; the group owns one worker thread per further member (queueing eight devices' launches from one
 * thread would take longer than the evaluation runs).  Not available on a group: the split-phase interface, caller-owned
 * buffers / streams, cmx_comm_attach*, cmx_backend_eval_many.  cmx_backend_set_window_from works on a group when the store was
 * created with cmx_events_create_group over the same devices.
 *   transport: CMX_GROUP_RCCL  -- ncclCommInitAll, one communicator per member (devices must be distinct);
 *              CMX_GROUP_DIRECT -- peer-to-peer reduce-scatter + all-gather kernels over the members' own buffers, ordered by
 *                                  HIP events (needs peer access between the devices; the only form for members that share
 *                                  ONE device, which is how a single-GPU box exercises all of this);
 *              CMX_GROUP_AUTO   -- MEASURED: every transport the devices allow is set up (members sharing a device: DIRECT only;
 *                                  no peer access: RCCL only), a production-sized exchange is timed through each at creation and
 *                                  the faster one kept (cmx_group_transport_info reports the choice and both timings).
 * n_devices == 1 returns a plain context (no group, no overhead).  Results equal the single-context evaluation of the whole
 * window to summation order (the planes are sums of the members' partial planes). */
enum { CMX_GROUP_AUTO = 0, CMX_GROUP_RCCL = 1, CMX_GROUP_DIRECT = 2 };
int cmx_backend_create_group(cmx_ctx **out, const int *devices, int n_devices, int W, int H, const double *lut, int Wp, int Hp,
                             int transport);
/* what a handle is made of: members, their devices, the transport in use, packed events per member of the current window,
 * host microseconds of the last fan-out (command published -> all members returned).  Any pointer may be NULL. */
int cmx_group_info(cmx_ctx *ctx, int *n_members, int *devices, int max_devices, int *transport, int64_t *events_per_member,
                   double *last_fanout_us);
/* how the transport in use was picked: *chosen = CMX_GROUP_RCCL / _DIRECT; *measured = 1 when CMX_GROUP_AUTO timed the candidates at
 * creation (20 staged exchanges of a 1 MB message -- a production tile set -- through every transport the devices allow, each member
 * on its own thread as an evaluation issues them) and kept the faster; *us_direct / *us_rccl = microseconds per exchange (-1: that
 * transport could not be set up on these devices -- e.g. RCCL for members sharing one device -- or was not timed).  Any pointer may
 * be NULL; a plain context reports CMX_GROUP_AUTO, 0, -1, -1. */
int cmx_group_transport_info(cmx_ctx *ctx, int *chosen, int *measured, double *us_direct, double *us_rccl);

/* ------------------------------------------------------------------ optimiser driver (host C++) ----------
 * The reference runs GSL's Fletcher-Reeves conjugate gradient around the cost functors
 * (src/frontend/local_optim_contrast_gsl.cpp:74-233, src/backend/global_optim_contrast_gsl.cpp:15-145).  These
 * entry points restate those driver loops (same constants and stopping rules) over a restated conjugate_fr
 * (GSL itself is not vendored by the reference); a host that keeps GSL simply does not call them. */
typedef struct {
  int iterations;      /* gsl_multimin_fdfminimizer_iterate calls (line searches) */
  int status;          /* 0 = converged on an accepted step, -2 = iteration limit, 27 = no progress (GSL codes) */
  int n_f, n_df;       /* cost-only and cost+gradient evaluations performed */
  double initial_cost; /* -contrast at the start */
  double final_cost;   /* -contrast at the returned parameters */
} cmx_solve_report;
/* ang_vel: in = warm start (the reference keeps ang_vel_ between packets), out = estimate */
int cmx_frontend_solve(cmx_ctx *ctx, double ang_vel[3], cmx_solve_report *report);
/* drotv: in = start (the reference uses 0), out = optimal incremental rotation vectors, n_params = 3*(K-num_fixed) */
int cmx_backend_solve(cmx_ctx *ctx, int n_params, double *drotv, cmx_solve_report *report);
/* the same driver loop over an arbitrary functor triple (the shape of gsl_multimin_function_fdf); lets a host or a
 * test run the identical optimiser over any other implementation of the cost */
typedef double (*cmx_f_fn)(const double *x, void *params);
typedef void (*cmx_df_fn)(const double *x, void *params, double *g);
typedef void (*cmx_fdf_fn)(const double *x, void *params, double *f, double *g);
int cmx_frcg_minimize(cmx_f_fn f, cmx_df_fn df, cmx_fdf_fn fdf, void *params, int n, double *x, double step_size,
                      double tol, double epsabs_grad, double tolfun, int max_iterations, cmx_solve_report *report);
/* the same with a fourth callback that receives, in front of every cost-only evaluation, the line search's acceptance test
 * (threshold, mode: see cmx_hint_next_df) -- a host that keeps its own f / df / fdf bodies forwards it to cmx_hint_next_df
 * and gets the gated gradient pass; hint == NULL is cmx_frcg_minimize */
typedef void (*cmx_hint_fn)(double threshold, int mode, void *params);
int cmx_frcg_minimize_hinted(cmx_f_fn f, cmx_df_fn df, cmx_fdf_fn fdf, cmx_hint_fn hint, void *params, int n, double *x,
                             double step_size, double tol, double epsabs_grad, double tolfun, int max_iterations,
                             cmx_solve_report *report);

/* ------------------------------------------------------------------ control-pose initialisation (host C++) ---
 * what the back-end thread does between two window solves to turn the front end's
 * angular velocities into the control poses the next window starts from, and the once-per-camera bearing table.
 * Pure host fp64 (tens of poses x a handful of control poses); no context, no device work.  Quaternions are
 * (x,y,z,w); stamps are int64 ns; sequences must be strictly increasing in time (the reference keeps them in
 * std::map keyed by stamp). */
/* PoseGraphOptimizer::integrateAngVel (src/backend/pose_graph_optimizer.cpp:191-222): trapezoidal integration of the
 * n stamped angular velocities onto (pose_t_ns, pose_quat), post-multiplied.  prev_*: ang_vel_prev_ (in/out).
 * out_*: room for n poses; *n_out = number written (stale stamps are skipped unless first_time_window). */
int cmx_integrate_ang_vel(int n, const int64_t *t_ns, const double *ang_vel /* 3n */, int64_t pose_t_ns,
                          const double pose_quat[4], int64_t *prev_t_ns, double prev_ang_vel[3], int first_time_window,
                          int64_t *out_t_ns, double *out_quat /* 4n */, int *n_out);
/* {Linear,Cubic}Trajectory::generateCtrlPoses' count (src/backend/trajectory.cpp:205-214 / :480-489):
 * round((t_end - t_beg).toSec() / dt_knots) + 1 (order 2) or + 3 (order 4); -1 on bad arguments */
int cmx_num_ctrl_poses(int order, int64_t t_beg_ns, int64_t t_end_ns, double dt_knots);
/* {Linear,Cubic}Trajectory::fitCtrlPoses (src/backend/trajectory.cpp:112-192 / :357-464): least-squares B-spline
 * fit in the tangent space at the first pose, solved like Eigen's fullPivHouseholderQr().solve (zero free variables
 * when rank-deficient).  CMX_ERR_INVALID_ARG where the reference's CHECK_GE / Eigen index assertion would abort. */
int cmx_fit_ctrl_poses(int order, int n_poses, const int64_t *t_ns, const double *quat /* 4 n_poses */,
                       double t_beg_sec, double dt_knots, int num_cps, double *out_quat /* 4 num_cps */);
/* {Linear,Cubic}Trajectory::incrementalUpdate (src/backend/trajectory.cpp:221-238 / :491-499):
 * knot_i <- exp(drotv_{i-idx_beg}) * knot_i for i >= idx_beg; n_params = 3*(K-idx_beg) (what cmx_backend_solve returns) */
int cmx_traj_incremental_update(int K, double *knots /* 4K, in/out */, int idx_beg, int n_params, const double *drotv);
/* {Linear,Cubic}Trajectory::evaluate without the Jacobian (src/backend/trajectory.cpp:86-110 / :329-355) */
int cmx_traj_evaluate(int order, int K, const double *knots, int64_t start_ns, int64_t dt_ns, int64_t t_ns,
                      double quat_out[4]);
/* CMaxSLAM::precomputeBearingVectors (src/cmax_slam.cpp:106-120): lut[(y*W+x)*3..] = projectPixelTo3dRay(
 * rectifyPoint(x,y)) of image_geometry's pinhole model, plumb_bob D = k1 k2 p1 p2 k3.  K 3x3, R 3x3, P 3x4
 * row-major; D, R, P may be NULL (no distortion, identity, [K|0]).  image_geometry / cv::undistortPoints are not
 * vendored by the reference: restated from their documented behaviour, parity unpinned. */
int cmx_bearing_lut(int W, int H, const double K[9], const double D[5], const double R[9], const double P[12],
                    double *lut /* W*H*3 */);

/* ------------------------------------------------------------------ timing hooks ------------------------
 * HIP-event timing of the dominant kernels on the context's stream (bench.py's roofline leg).
 * cmx_timing_enable(ctx, mask) makes every evaluation record events around the kernel classes whose bit is set
 * (bit CMX_T_SPLAT, ...; 0xff = all, 0 = off); mask | (n << 8) samples every n-th evaluation only.  The per-event
 * kernels (splat, gather) carry their events on the kernel itself (hipExtLaunchKernelGGL start / stop: the dispatch's
 * own timestamps, the same rocprofv3 reports); the other classes are bracketed on the stream (CMX_T_COMM: the RCCL collectives of an attached communicator,
 * i.e. including the wait for the slowest rank);
 * cmx_timing_get returns accumulated milliseconds and launch counts per kernel class, then resets. */
enum { CMX_T_SPLAT = 0, CMX_T_IMAGE = 1, CMX_T_POSE = 2, CMX_T_GATHER = 3, CMX_T_ZERO = 4, CMX_T_COMM = 5, CMX_T_FINAL = 6, CMX_T_BATCH = 7, CMX_T_COUNT = 8 };
/* CMX_T_FINAL: the separate finalize launch (absent when CMX_OPT_TAIL_FINALIZE folds it into the last kernel);
 * CMX_T_BATCH: the back end's per-batch pass of the gradient gather.  Every class except CMX_T_ZERO / CMX_T_COMM is timed
 * through events carried by its (main) kernel: the dispatch's own begin / end timestamps, what rocprofv3 reports. */
/* cmx_get_stats: counters of a context, one double each, indexed by this enum (ABI 6: the indices have names; 0..16 keep the
 * values they had as bare numbers). */
enum {
  CMX_STAT_REBINS = 0,              /* destination-tile sorts so far */
  CMX_STAT_FALLBACK_FRAC = 1,       /* fraction of votes that left their LDS window in the last evaluation */
  CMX_STAT_CHUNKS = 2,              /* workgroup chunks of the current sort */
  CMX_STAT_EVENTS = 3,              /* packed (sub-sampled) events */
  CMX_STAT_REUSE_HITS = 4,          /* gradient evaluations that reused the resident image of the previous cost evaluation */
  CMX_STAT_SHARDED_HOST_SYNCS = 5,  /* host synchronisations between the splat and the last kernel of sharded evaluations (stays 0) */
  CMX_STAT_EXCHANGE_MISSES = 6,     /* sharded evaluations whose exchange set missed flagged tiles (completed by a second exchange) */
  CMX_STAT_EXCHANGE_TILES = 7,      /* tiles in the current exchange set (-1: none known, whole planes) */
  CMX_STAT_COMM_BYTES = 8,          /* bytes the last sharded evaluation exchanged (all collectives, this rank's buffers) */
  CMX_STAT_SPEC_IMAGES = 9,         /* cost-only evaluations that ran the adjoint image pass speculatively */
  CMX_STAT_SPEC_HITS = 10,          /* gradient evaluations that found it ready */
  CMX_STAT_GATED_LAUNCHES = 11,     /* gated gradient passes queued (cmx_hint_next_df) */
  CMX_STAT_GATED_HITS = 12,         /* gradient evaluations served by one */
  CMX_STAT_CHAIN_SOLVES = 13,       /* device-driven solves started */
  CMX_STAT_CHAIN_SLOTS = 14,        /* evaluation slots they queued */
  CMX_STAT_CHAIN_TAKEOVERS = 15,    /* solves the host took over (disagreement, or votes outside their windows in a fused slot) */
  CMX_STAT_CHAIN_WARM_STARTS = 16,  /* device-driven solves that started warm (nothing copied or cleared in front of them) */
  CMX_STAT_FUSED_EVALS = 17,        /* evaluations whose image pass ran inside the splat launch (two launches instead of three) */
  CMX_STAT_FUSED_REDOS = 18,        /* ... of which were repeated (votes beyond the reach of their tiles' arrival counts) */
  CMX_STAT_ONE_LAUNCH_EVALS = 19,   /* ... of which ran splat, image pass, gather and finalize as ONE launch */
  CMX_STAT_FUSED_TIMEOUTS = 20,     /* repeats caused by a tile workgroup that gave up waiting (stays 0: a wait is bounded at 2 ms) */
  CMX_STAT_SELF_SERVE_EVALS = 21,   /* ... of the one-launch evaluations, those of the self-service form (chunk workgroups alone) */
  CMX_N_STATS = 22
};
int cmx_get_stats(cmx_ctx *ctx, double *stats, int n_stats); /* writes min(n_stats, CMX_N_STATS) entries (ABI 3: the length is explicit) */
/* ABI revision of this header: bumped whenever a signature or the layout of a caller-provided buffer changes
 * (3: cmx_get_stats takes the buffer length; cmx_frontend_prepare / cmx_backend_prepare added;
 *  4: groups, cmx_backend_get_pose_table, stream priority / CU mask;
 *  5: cmx_comm_info; the event store behind a group (cmx_events_create_group, cmx_backend_set_window_from on a group);
 *     CMX_OPT_SPIN_WAIT values >= 2 are a spin budget in microseconds for all three waiters (before: "spin"), negative values are
 *     rejected (before: accepted as non-zero);
 *  6: named cmx_get_stats indices, five more of them (the buffer length is the caller's: old callers keep reading what they asked for); cmx_group_transport_info; the *_aos entry points; cmx_set_stream_priority /
 *     cmx_set_cu_mask moved to cmax_hip_diag.h) */
#define CMX_ABI_VERSION 6
int cmx_abi_version(void);
int cmx_timing_enable(cmx_ctx *ctx, int on);
int cmx_timing_get(cmx_ctx *ctx, double ms[CMX_T_COUNT], int64_t launches[CMX_T_COUNT]);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CMAX_HIP_H */
