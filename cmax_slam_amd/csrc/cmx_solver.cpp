// cmx_solver.cpp -- the reference's optimiser driver loops over the HIP evaluator (C ABI entry points
// cmx_frontend_solve / cmx_backend_solve).  Restates
//   AngVelEstimator::setupProblemAndOptimize_gsl      src/frontend/local_optim_contrast_gsl.cpp:74-233
//   PoseGraphOptimizer::setupProblemAndOptimize_gsl   src/backend/global_optim_contrast_gsl.cpp:15-145
// around the FR-CG state machine (restated GSL conjugate_fr + the loops above, cmx_frcg_sm.hpp).  The cost functors below are exactly the
// bodies INTEGRATION.md gives for local_contrast_{f,df,fdf} / global_contrast_{f,df,fdf}.
#include <math.h>

#include <vector>

#include "cmx_context.hpp"

namespace {

struct SolveState {
  cmx_ctx *ctx;
  bool backend;
  int n;
  int err = CMX_OK;
  std::vector<double> g;
};

int eval(SolveState *s, const double *x, double *contrast, double *grad) {
  const int rc = s->backend ? cmx_backend_eval(s->ctx, x, contrast, grad) : cmx_frontend_eval(s->ctx, x, contrast, grad);
  if (rc != CMX_OK && s->err == CMX_OK) s->err = rc;
  return rc;
}

// *_contrast_fdf: f = -contrast, df = -gradient (minimise -contrast); a failed evaluation maps to NaN
void contrast_fdf(const double *x, void *p, double *f, double *df) {
  SolveState *s = static_cast<SolveState *>(p);
  double c = 0;
  if (eval(s, x, &c, df ? s->g.data() : nullptr) != CMX_OK) {
    *f = NAN;
    if (df) for (int i = 0; i < s->n; i++) df[i] = NAN;
    return;
  }
  *f = -c;
  if (df) for (int i = 0; i < s->n; i++) df[i] = -s->g[i];
}
double contrast_f(const double *x, void *p) {
  double cost;
  contrast_fdf(x, p, &cost, nullptr);
  return cost;
}
void contrast_df(const double *x, void *p, double *df) {
  double cost;
  contrast_fdf(x, p, &cost, df);
}
void contrast_hint(double thr, int mode, void *p) {  // the line search's acceptance test, handed to the evaluator
  SolveState *s = static_cast<SolveState *>(p);
  (void)cmx_hint_next_df(s->ctx, thr, mode);
}

// the FR-CG state machine of cmx_frcg_sm.hpp with host storage, fed from callbacks one request at a time: the driver loop
// shared by both ends (and by cmx_frcg_minimize for arbitrary functors)
struct Machine {
  std::vector<double> store, g;
  cmx::FrcgSM s{};
  Machine(int n, const double *x0, double step, double tol, double epsabs_grad, double tolfun, int max_iter)
      : store((size_t)9 * n, 0.0), g((size_t)n, 0.0) {
    double *v = store.data();  // one contiguous block in the order cmx_chain.cpp / the device's finalize step expect
    s.x = v; s.gradient = v + n; s.dx = v + 2 * n; s.x1 = v + 3 * n; s.dx1 = v + 4 * n; s.x2 = v + 5 * n; s.dx2 = v + 6 * n;
    s.p = v + 7 * n; s.g0 = v + 8 * n;
    for (int i = 0; i < n; i++) s.x[i] = x0[i];
    cmx::sm_begin(s, n, step, tol, epsabs_grad, tolfun, max_iter);
  }
  void run_host(const cmx::FunctionFdf &fn, const int *err) {
    // a machine handed back in the middle of a point (the device-driven chain stopped between a cost and its gradient):
    // the gradient at that very point comes first
    if (s.phase == cmx::SM_TRIAL_G || s.phase == cmx::SM_IP_G || s.phase == cmx::SM_MIN_G) {
      fn.df(cmx::sm_point(s), fn.params, g.data());
      cmx::sm_grad(s, g.data());
    }
    while (!cmx::sm_done(s) && (!err || *err == CMX_OK)) cmx::sm_step_host(s, fn, g.data());
  }
  void report(double *x_out, cmx_solve_report *rep) const {
    for (int i = 0; i < s.n; i++) x_out[i] = s.x[i];
    if (rep) {
      rep->iterations = s.iter;
      rep->status = s.status;
      rep->n_f = s.n_f;
      rep->n_df = s.n_df;
      rep->initial_cost = s.initial_cost;
      rep->final_cost = s.f;
    }
  }
};

void drive(const cmx::FunctionFdf &user, double *x_inout, double initial_step_size, double tol, double epsabs_grad,
           double tolfun, int num_max_line_searches, cmx_solve_report *rep) {
  Machine m((int)user.n, x_inout, initial_step_size, tol, epsabs_grad, tolfun, num_max_line_searches);
  m.run_host(user, nullptr);  // the first request is gsl_multimin_fdfminimizer_set's fdf: "This call already evaluates the function"
  m.report(x_inout, rep);
}

int solve(cmx_ctx *ctx, bool backend, int n, double *x_inout, double tol, double epsabs_grad, cmx_solve_report *rep) {
  UrgentScope urgent(ctx);  // the whole solve is one burst
  SolveState st;
  st.ctx = ctx;
  st.backend = backend;
  st.n = n;
  st.g.assign((size_t)(n > 0 ? n : 1), 0.0);
  cmx::FunctionFdf fn{contrast_f, contrast_df, contrast_fdf, (size_t)n, &st};
  fn.hint = contrast_hint;
  Machine m(n, x_inout, 0.1, tol, epsabs_grad, 1e-4, 50);
  bool completed = false;
  if (!backend) {  // front end: the line search runs ahead of the host on the device as far as it goes (cmx_chain.cpp)
    const int rc = chain_run_frontend(ctx, m.s, &completed);
    if (rc != CMX_OK) return rc;
  }
  if (!completed) m.run_host(fn, &st.err);
  m.report(x_inout, rep);
  return st.err;
}

}  // namespace

extern "C" {

int cmx_frontend_solve(cmx_ctx *ctx, double ang_vel[3], cmx_solve_report *report) {
  if (!ctx || !ang_vel) return CMX_ERR_INVALID_ARG;
  // step 0.1, tol 0.05, <= 50 line searches, |g| < 1e-3, |1 - c_new/c_old| < 1e-4   (:106-122)
  return solve(ctx, false, 3, ang_vel, 0.05, 1e-3, report);
}

int cmx_backend_solve(cmx_ctx *ctx, int n_params, double *drotv, cmx_solve_report *report) {
  if (!ctx || (n_params > 0 && !drotv) || n_params < 0) return CMX_ERR_INVALID_ARG;
  // x0 is whatever the caller passes (the reference starts at 0, global_optim_contrast_gsl.cpp:37);
  // step 0.1, tol 0.1, <= 50 line searches, |g| < 1e-4, tolfun 1e-4   (:41-53)
  return solve(ctx, true, n_params, drotv, 0.1, 1e-4, report);
}

int cmx_frcg_minimize(cmx_f_fn f, cmx_df_fn df, cmx_fdf_fn fdf, void *params, int n, double *x, double step_size,
                      double tol, double epsabs_grad, double tolfun, int max_iterations, cmx_solve_report *report) {
  if (!f || !df || !fdf || n <= 0 || !x) return CMX_ERR_INVALID_ARG;
  cmx::FunctionFdf fn{f, df, fdf, (size_t)n, params};
  drive(fn, x, step_size, tol, epsabs_grad, tolfun, max_iterations, report);
  return CMX_OK;
}

int cmx_frcg_minimize_hinted(cmx_f_fn f, cmx_df_fn df, cmx_fdf_fn fdf, cmx_hint_fn hint, void *params, int n, double *x,
                             double step_size, double tol, double epsabs_grad, double tolfun, int max_iterations,
                             cmx_solve_report *report) {
  if (!f || !df || !fdf || n <= 0 || !x) return CMX_ERR_INVALID_ARG;
  cmx::FunctionFdf fn{f, df, fdf, (size_t)n, params};
  fn.hint = hint;
  drive(fn, x, step_size, tol, epsabs_grad, tolfun, max_iterations, report);
  return CMX_OK;
}

}  // extern "C"
