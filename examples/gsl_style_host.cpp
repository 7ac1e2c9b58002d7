// gsl_style_host.cpp -- a plain C++ host driving libcmaxhip.so exactly the way the reference's GSL glue would:
// a params struct, the three callbacks f / df / fdf with gsl_multimin_function_fdf's shape, and a minimiser that only
// sees those callbacks (here: cmx_frcg_minimize, the restated conjugate_fr; with GSL present the same three functions
// are what gsl_multimin_fdfminimizer_set receives -- see INTEGRATION.md).  No Python, no torch.
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

namespace {

struct Estimator {  // stands in for cmax_slam::AngVelEstimator: owns the evaluator context
  cmx_ctx *cmx = nullptr;
  int status = CMX_OK;
};

// local_contrast_fdf (src/frontend/local_optim_contrast_gsl.cpp:20-56) with its new body
void local_contrast_fdf(const double *v, void *ptr, double *f, double *df) {
  Estimator *est = static_cast<Estimator *>(ptr);
  double contrast = 0, g[3];
  const int rc = cmx_frontend_eval(est->cmx, v, &contrast, df ? g : nullptr);
  if (rc != CMX_OK) {
    est->status = rc;
    *f = NAN;
    return;
  }
  *f = -contrast;
  if (df)
    for (int i = 0; i < 3; i++) df[i] = -g[i];
}
double local_contrast_f(const double *v, void *p) {
  double cost;
  local_contrast_fdf(v, p, &cost, nullptr);
  return cost;
}
void local_contrast_df(const double *v, void *p, double *df) {
  double cost;
  local_contrast_fdf(v, p, &cost, df);
}
// not in GSL's gsl_multimin_function_fdf: the line search's acceptance test, forwarded to the evaluator so that the gradient
// pass is queued (gated on the device) behind the cost evaluation it will follow ("Line-search hint")
void local_contrast_hint(double threshold, int mode, void *p) {
  cmx_hint_next_df(static_cast<Estimator *>(p)->cmx, threshold, mode);
}

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
  rc = cmx_frontend_set_packet(est.cmx, n, x.data(), y.data(), t.data(), t_ref, K[0], K[1], K[2], K[3], 100, 1.0, CMX_VARIANCE);
  if (rc != CMX_OK) {
    fprintf(stderr, "set_packet failed: %s: %s\n", cmx_status_string(rc), cmx_last_error(est.cmx));
    return 1;
  }
  // one plain evaluation through the callback, then the solve with the reference's constants
  double w0[3] = {0.3, -0.5, 0.2}, f0, g0[3];
  local_contrast_fdf(w0, &est, &f0, g0);
  double w[3] = {0, 0, 0};  // ang_vel_ starts at 0 (ang_vel_estimator.cpp:26)
  cmx_solve_report rep;
  rc = cmx_frcg_minimize_hinted(local_contrast_f, local_contrast_df, local_contrast_fdf, local_contrast_hint, &est, 3, w, 0.1, 0.05,
                                1e-3, 1e-4, 50, &rep);
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
