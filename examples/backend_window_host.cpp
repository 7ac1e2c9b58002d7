// backend_window_host.cpp -- one back-end time window driven from plain C++ through the C ABI, in the order
// PoseGraphOptimizer::processTimeWindow (src/backend/pose_graph_optimizer.cpp:244-323) does it:
//
//   integrateAngVel -> generateCtrlPoses (fit) -> window hand-over -> GSL-shaped f / df / fdf solve ->
//   incrementalUpdate -> updateIG -> setUpdateTimesIG -> pose_latest for the next window
//
// with gsl_multimin_function_fdf-shaped callbacks (global_contrast_f / _df / _fdf,
// src/backend/global_optim_contrast_gsl_analytical.cpp:17-81) over cmx_frcg_minimize.  No Python, no torch, no Eigen.
//
//   run:  examples/backend_window_host window.bin [devices] [store]   (window.bin written by tests/test_gpu_cpp_host.py)
//
// devices (optional, e.g. "0,1,2,3" or "0,0"): the same host code on a GROUP handle (cmx_backend_create_group) -- one
// process, one thread, one optimiser, the window's batches sharded over the listed devices; nothing else in this file changes.
// store (optional, the literal word): the events go through the device-resident event store the way the reference keeps them in
// AngVelEstimator::events_ -- pushed in chunks as they "arrive" (cmx_events_push; with a device list: cmx_events_create_group, one
// replica per device), the window CUT from it by global index (cmx_backend_set_window_from = PoseGraphOptimizer::getEventSubset,
// pose_graph_optimizer.cpp:131-165, without the copy; on a group every member cuts its own batch range on its own device), the
// events before the next window's start dropped afterwards (cmx_events_drop_before = deleteOldEvents, ang_vel_estimator.cpp:149-173).
//
// window.bin (little endian): int32 W, H, Wp, Hp, order, K, num_fixed, n_av; int64 n, start_ns, dt_ns, t_next_ns,
//   t_win_beg_ns, t_win_end_ns; double dt_knots; uint16 x[n], y[n]; int64 t_ns[n]; double lut[W*H*3];
//   double knots[4K] (the window's control poses as the front end initialised them); int64 av_t[n_av]; double av_w[3 n_av]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cmax_hip.h"

namespace {

struct Optimizer {  // stands in for cmax_slam::PoseGraphOptimizer
  cmx_ctx *cmx = nullptr;
  int n_params = 0;
  int status = CMX_OK;
};

// global_contrast_fdf with its new body: the left-multiplicative knot update happens inside cmx_backend_eval
void global_contrast_fdf(const double *v, void *ptr, double *f, double *df) {
  Optimizer *opt = static_cast<Optimizer *>(ptr);
  double contrast = 0;
  std::vector<double> g(opt->n_params);
  const int rc = cmx_backend_eval(opt->cmx, v, &contrast, df ? g.data() : nullptr);
  if (rc != CMX_OK) {
    opt->status = rc;
    *f = NAN;
    return;
  }
  *f = -contrast;
  if (df)
    for (int i = 0; i < opt->n_params; i++) df[i] = -g[i];
}
double global_contrast_f(const double *v, void *p) {
  double cost;
  global_contrast_fdf(v, p, &cost, nullptr);
  return cost;
}
void global_contrast_df(const double *v, void *p, double *df) {
  double cost;
  global_contrast_fdf(v, p, &cost, df);
}

template <typename T>
bool read_vec(FILE *fp, std::vector<T> &v, size_t n) {
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, fp) == n;
}
#define CHECK_RC(call)                                                                                   \
  do {                                                                                                   \
    const int rc_ = (call);                                                                              \
    if (rc_ != CMX_OK) {                                                                                 \
      fprintf(stderr, "%s failed: %s: %s\n", #call, cmx_status_string(rc_), cmx_last_error(opt.cmx));    \
      return 1;                                                                                          \
    }                                                                                                    \
  } while (0)

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s window.bin [devices, e.g. 0,1] [store]\n", argv[0]);
    return 2;
  }
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  int32_t hdr[8];
  int64_t big[6];
  double dt_knots;
  if (fread(hdr, 4, 8, fp) != 8 || fread(big, 8, 6, fp) != 6 || fread(&dt_knots, 8, 1, fp) != 1) return 2;
  const int W = hdr[0], H = hdr[1], Wp = hdr[2], Hp = hdr[3], order = hdr[4], K = hdr[5], num_fixed = hdr[6], n_av = hdr[7];
  const int64_t n = big[0], start_ns = big[1], dt_ns = big[2], t_next_ns = big[3], t_win_beg = big[4], t_win_end = big[5];
  std::vector<uint16_t> x, y;
  std::vector<int64_t> t, av_t;
  std::vector<double> lut, knots, av_w;
  if (!read_vec(fp, x, (size_t)n) || !read_vec(fp, y, (size_t)n) || !read_vec(fp, t, (size_t)n) ||
      !read_vec(fp, lut, (size_t)W * H * 3) || !read_vec(fp, knots, (size_t)4 * K) || !read_vec(fp, av_t, (size_t)n_av) ||
      !read_vec(fp, av_w, (size_t)3 * n_av))
    return 2;
  fclose(fp);

  Optimizer opt;
  opt.n_params = 3 * (K - num_fixed);

  // --- control-pose initialisation from the front end's angular velocities (host fp64, no device work)
  std::vector<int64_t> pose_t((size_t)n_av);
  std::vector<double> pose_q((size_t)4 * n_av), fitted;
  int n_poses = 0, n_cp = 0;
  if (n_av > 0) {
    int64_t prev_t = av_t[0];
    double prev_w[3] = {av_w[0], av_w[1], av_w[2]};
    const double q0[4] = {knots[0], knots[1], knots[2], knots[3]};  // pose_latest_: the window starts at its first knot
    if (cmx_integrate_ang_vel(n_av, av_t.data(), av_w.data(), t_win_beg, q0, &prev_t, prev_w, /*first window*/ 1,
                              pose_t.data(), pose_q.data(), &n_poses) != CMX_OK)
      return 1;
    n_cp = cmx_num_ctrl_poses(order, t_win_beg, t_win_end, dt_knots);
    fitted.resize((size_t)4 * (n_cp > 0 ? n_cp : 1));
    const double t_beg_sec = (double)(t_win_beg / 1000000000LL) + 1e-9 * (double)(t_win_beg % 1000000000LL);
    if (n_cp <= 0 || cmx_fit_ctrl_poses(order, n_poses, pose_t.data(), pose_q.data(), t_beg_sec, dt_knots, n_cp, fitted.data()) != CMX_OK)
      return 1;
  }

  // --- the window solve on the device
  std::vector<int> devices;
  if (argc > 2)
    for (const char *p = argv[2]; *p;) {
      char *end;
      devices.push_back((int)strtol(p, &end, 10));
      p = (*end == ',') ? end + 1 : end;
      if (end == p && *end) break;
    }
  if (devices.size() > 1) CHECK_RC(cmx_backend_create_group(&opt.cmx, devices.data(), (int)devices.size(), W, H, lut.data(), Wp, Hp, CMX_GROUP_AUTO));
  else CHECK_RC(cmx_backend_create(&opt.cmx, devices.empty() ? 0 : devices[0], W, H, lut.data(), Wp, Hp));
  const bool use_store = argc > 3 && std::string(argv[3]) == "store";
  cmx_events *store = nullptr;
  if (use_store) {
    const int dev0 = devices.empty() ? 0 : devices[0];
    if (devices.size() > 1) CHECK_RC(cmx_events_create_group(&store, devices.data(), (int)devices.size(), W, H, (size_t)n + 1024));
    else CHECK_RC(cmx_events_create(&store, dev0, W, H, (size_t)n + 1024));
    const int64_t chunk = 7000;  // the stream arrives in packets
    for (int64_t at = 0; at < n; at += chunk) {
      const int64_t m = (n - at < chunk) ? n - at : chunk;
      if (cmx_events_push(store, m, x.data() + at, y.data() + at, t.data() + at) != CMX_OK) {
        fprintf(stderr, "cmx_events_push: %s\n", cmx_events_last_error(store));
        return 1;
      }
    }
    CHECK_RC(cmx_backend_set_window_from(opt.cmx, store, cmx_events_begin(store), n, order, K, knots.data(), start_ns, dt_ns, num_fixed,
                                         t_next_ns, 100, 1, 1.0, CMX_VARIANCE, CMX_KEEP_MAP));
  } else
  CHECK_RC(cmx_backend_set_window(opt.cmx, n, x.data(), y.data(), t.data(), order, K, knots.data(), start_ns, dt_ns, num_fixed,
                                  t_next_ns, 100, 1, 1.0, CMX_VARIANCE, CMX_KEEP_MAP));  // the global map stays on the GPU
  std::vector<double> v0((size_t)opt.n_params, 0.0), g0((size_t)opt.n_params), drotv((size_t)opt.n_params, 0.0);
  double f0;
  global_contrast_fdf(v0.data(), &opt, &f0, g0.data());
  cmx_solve_report rep;
  int rc = cmx_frcg_minimize(global_contrast_f, global_contrast_df, global_contrast_fdf, &opt, opt.n_params, drotv.data(), 0.1,
                             0.1, 1e-4, 1e-4, 50, &rep);  // global_optim_contrast_gsl.cpp:41-53
  if (rc != CMX_OK || opt.status != CMX_OK) {
    fprintf(stderr, "solve failed\n");
    return 1;
  }
  // --- after the solve: traj_->incrementalUpdate, updateIG, setUpdateTimesIG, pose_latest_
  CHECK_RC(cmx_traj_incremental_update(K, knots.data(), num_fixed, opt.n_params, drotv.data()));
  CHECK_RC(cmx_backend_update_map(opt.cmx, /*max_update_times*/ 200));
  int marked = 0;
  for (int64_t tc = t_win_beg; tc < t_next_ns; tc += 50000000LL, marked++) {  // every 0.05 s over the stride
    double q[4];
    CHECK_RC(cmx_traj_evaluate(order, K, knots.data(), start_ns, dt_ns, tc, q));
    CHECK_RC(cmx_backend_mark_visited(opt.cmx, q, 3));
  }
  double q_latest[4];
  CHECK_RC(cmx_traj_evaluate(order, K, knots.data(), start_ns, dt_ns, t_win_end - 1000, q_latest));  // t_win_end - 1e-6 s
  std::vector<float> map((size_t)Wp * Hp);
  std::vector<unsigned char> visits((size_t)Wp * Hp);
  CHECK_RC(cmx_backend_get_map(opt.cmx, map.data(), visits.data()));
  double map_sum = 0;
  long visited = 0;
  for (size_t i = 0; i < map.size(); i++) { map_sum += map[i]; visited += visits[i] != 0; }

  printf("f0 %.17g\ng0", f0);
  for (double g : g0) printf(" %.17g", g);
  printf("\ndrotv");
  for (double d : drotv) printf(" %.17g", d);
  printf("\nknots");
  for (double k : knots) printf(" %.17g", k);
  printf("\niterations %d n_f %d n_df %d initial %.17g final %.17g\n", rep.iterations, rep.n_f, rep.n_df, rep.initial_cost,
         rep.final_cost);
  printf("fitted");
  for (double k : fitted) printf(" %.17g", k);
  printf("\nlatest %.17g %.17g %.17g %.17g\nmap %.17g %ld %d\n", q_latest[0], q_latest[1], q_latest[2], q_latest[3], map_sum,
         visited, marked);
  if (store) {  // deleteOldEvents: everything before the next window's first event
    int64_t keep_from = 0;
    while (keep_from < n && t[(size_t)keep_from] < t_next_ns) keep_from++;
    if (cmx_events_drop_before(store, cmx_events_begin(store) + keep_from) != CMX_OK) return 1;
    printf("store %lld %lld\n", (long long)cmx_events_begin(store), (long long)cmx_events_end(store));
    cmx_events_destroy(store);
  }
  cmx_destroy(opt.cmx);
  return 0;
}
