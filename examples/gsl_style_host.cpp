// gsl_style_host.cpp -- a plain C++ host driving libcmaxhip.so exactly the way the reference's GSL glue does: a params
// object, the three callbacks f / df / fdf WITH THE REFERENCE'S SIGNATURES (gsl_vector in, gsl_vector out:
// src/frontend/local_optim_contrast_gsl.cpp:19-70), a gsl_multimin_function_fdf filled like :87-96 -- and a minimiser that only
// sees that struct.  Here the minimiser is cmx_frcg_minimize_hinted (the restated conjugate_fr) behind a thin adapter; with GSL
// installed the SAME struct goes to gsl_multimin_fdfminimizer_set and examples/gsl_shim.h defines nothing.
// The events are handed over as the reference holds them: an array of dvs_msgs::Event records (cmx_frontend_set_packet_aos).
// No Python, no torch.
//
//   build:  g++ -std=c++17 -O2 -I include examples/gsl_style_host.cpp -o examples/gsl_style_host \
//               cmax_slam_amd/libcmaxhip.so -Wl,-rpath,'$ORIGIN/../cmax_slam_amd'
//   run:    examples/gsl_style_host events.bin      (binary file written by tests/test_gpu_cpp_host.py)
//
// events.bin layout (little endian): int32 W, H; int64 n; int64 t_ref_ns; double fx, fy, cx, cy;
//                                    uint16 x[n]; uint16 y[n]; int64 t_ns[n]; double lut[W*H*3]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "cmax_hip.h"
#include "gsl_shim.h"

namespace {

// dvs_msgs::Event as roscpp generates it: uint16 x, y; ros::Time ts {uint32 sec, nsec}; uint8 polarity (16 bytes)
struct RosTime { uint32_t sec, nsec; };
struct DvsEvent { uint16_t x, y; RosTime ts; uint8_t polarity; };
static_assert(sizeof(DvsEvent) == 16, "dvs_msgs::Event layout");

struct Estimator {  // stands in for cmax_slam::AngVelEstimator: owns the evaluator context
  cmx_ctx *cmx = nullptr;
  int status = CMX_OK;
};

// local_contrast_fdf (src/frontend/local_optim_contrast_gsl.cpp:19-56): the reference's signature, its new body
void local_contrast_fdf(const gsl_vector *v, void *ptr, double *f, gsl_vector *df) {
  Estimator *estimator = static_cast<Estimator *>(ptr);
  const double ang_vel[3] = {gsl_vector_get(v, 0), gsl_vector_get(v, 1), gsl_vector_get(v, 2)};
  double contrast = 0, gradient[3];
  const int rc = cmx_frontend_eval(estimator->cmx, ang_vel, &contrast, df ? gradient : nullptr);
  if (rc != CMX_OK) {
    estimator->status = rc;
    *f = NAN;  // GSL_NAN
    return;
  }
  *f = -contrast;  // change sign: minimize -contrast
  if (df != nullptr)
    for (int i = 0; i < 3; i++) gsl_vector_set(df, i, -gradient[i]);
}
double local_contrast_f(const gsl_vector *v, void *adata) {
  double cost;
  local_contrast_fdf(v, adata, &cost, nullptr);
  return cost;
}
void local_contrast_df(const gsl_vector *v, void *adata, gsl_vector *df) {
  double cost;
  local_contrast_fdf(v, adata, &cost, df);
}

// ---- adapter: a gsl_multimin_function_fdf in front of this library's pointer-based driver (what stands where
// gsl_multimin_fdfminimizer_set / _iterate stand in the reference, :98-215)
struct GslProblem {
  gsl_multimin_function_fdf *fn;
  Estimator *est;
};
double thunk_f(const double *x, void *p) {
  GslProblem *g = static_cast<GslProblem *>(p);
  gsl_vector v = cmx_gsl_view(const_cast<double *>(x), g->fn->n);
  return g->fn->f(&v, g->fn->params);
}
void thunk_df(const double *x, void *p, double *df) {
  GslProblem *g = static_cast<GslProblem *>(p);
  gsl_vector v = cmx_gsl_view(const_cast<double *>(x), g->fn->n), d = cmx_gsl_view(df, g->fn->n);
  g->fn->df(&v, g->fn->params, &d);
}
void thunk_fdf(const double *x, void *p, double *f, double *df) {
  GslProblem *g = static_cast<GslProblem *>(p);
  gsl_vector v = cmx_gsl_view(const_cast<double *>(x), g->fn->n), d = cmx_gsl_view(df, g->fn->n);
  g->fn->fdf(&v, g->fn->params, f, df ? &d : nullptr);
}
// not in GSL's gsl_multimin_function_fdf: the line search's acceptance test, forwarded to the evaluator so that the gradient
// pass is queued (gated on the device) behind the cost evaluation it will follow ("Line-search hint")
void thunk_hint(double threshold, int mode, void *p) { cmx_hint_next_df(static_cast<GslProblem *>(p)->est->cmx, threshold, mode); }

template <typename T>
bool read_vec(FILE *fp, std::vector<T> &v, size_t n) {
  v.resize(n);
  return fread(v.data(), sizeof(T), n, fp) == n;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s events.bin\n", argv[0]);
    return 2;
  }
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  int32_t W, H;
  int64_t n, t_ref;
  double K[4];
  if (fread(&W, 4, 1, fp) != 1 || fread(&H, 4, 1, fp) != 1 || fread(&n, 8, 1, fp) != 1 || fread(&t_ref, 8, 1, fp) != 1 ||
      fread(K, 8, 4, fp) != 4)
    return 2;
  std::vector<uint16_t> x, y;
  std::vector<int64_t> t;
  std::vector<double> lut;
  if (!read_vec(fp, x, (size_t)n) || !read_vec(fp, y, (size_t)n) || !read_vec(fp, t, (size_t)n) ||
      !read_vec(fp, lut, (size_t)W * H * 3))
    return 2;
  fclose(fp);

  Estimator est;
  int rc = cmx_frontend_create(&est.cmx, 0, W, H, lut.data());
  if (rc != CMX_OK) {
    fprintf(stderr, "create failed: %s\n", cmx_status_string(rc));
    return 1;
  }
  // (a new context already runs the production configuration: adjoint gradient + LDS-privatised splat)
  // the packet as the reference holds it -- std::vector<dvs_msgs::Event> event_subset_ (ang_vel_estimator.cpp:137-147) -- handed over
  // as it is: one packing pass inside the library, no x[] / y[] / t_ns[] vectors on the host's side
  std::vector<DvsEvent> event_subset((size_t)n);
  for (int64_t i = 0; i < n; i++)
    event_subset[(size_t)i] = DvsEvent{x[(size_t)i], y[(size_t)i], RosTime{(uint32_t)(t[(size_t)i] / 1000000000), (uint32_t)(t[(size_t)i] % 1000000000)},
                                        (uint8_t)(i & 1)};
  const cmx_aos_layout layout = CMX_AOS_DVS_EVENT;
  rc = cmx_frontend_set_packet_aos(est.cmx, n, event_subset.data(), &layout, t_ref, K[0], K[1], K[2], K[3], 100, 1.0, CMX_VARIANCE);
  if (rc != CMX_OK) {
    fprintf(stderr, "set_packet failed: %s: %s\n", cmx_status_string(rc), cmx_last_error(est.cmx));
    return 1;
  }
  // the solver's view of the problem, filled as the reference fills it (local_optim_contrast_gsl.cpp:87-96)
  gsl_multimin_function_fdf solver_info;
  solver_info.n = 3;
  solver_info.f = local_contrast_f;
  solver_info.df = local_contrast_df;
  solver_info.fdf = local_contrast_fdf;
  solver_info.params = &est;
  GslProblem prob{&solver_info, &est};
  // one plain evaluation through the callback, then the solve with the reference's constants
  double w0[3] = {0.3, -0.5, 0.2}, f0, g0[3];
  {
    gsl_vector v = cmx_gsl_view(w0, 3), d = cmx_gsl_view(g0, 3);
    solver_info.fdf(&v, solver_info.params, &f0, &d);
  }
  double w[3] = {0, 0, 0};  // ang_vel_ starts at 0 (ang_vel_estimator.cpp:26)
  cmx_solve_report rep;
  rc = cmx_frcg_minimize_hinted(thunk_f, thunk_df, thunk_fdf, thunk_hint, &prob, 3, w, 0.1, 0.05, 1e-3, 1e-4, 50, &rep);
  if (rc != CMX_OK || est.status != CMX_OK) {
    fprintf(stderr, "solve failed\n");
    return 1;
  }
  printf("f0 %.17g\ng0 %.17g %.17g %.17g\n", f0, g0[0], g0[1], g0[2]);
  printf("w %.17g %.17g %.17g\niterations %d n_f %d n_df %d initial %.17g final %.17g\n", w[0], w[1], w[2], rep.iterations,
         rep.n_f, rep.n_df, rep.initial_cost, rep.final_cost);
  cmx_destroy(est.cmx);
  return 0;
}
