// cmx_frcg_sm.hpp -- the Fletcher-Reeves conjugate-gradient minimiser AND the reference's driver loop as one resumable
// state machine: every call consumes the result of the evaluation it asked for last and returns with the next request.
//
// Why a state machine: the same code runs in two places.
//   host   (cmx_solver.cpp): run_host() feeds it from f / df / fdf callbacks -- the call pattern of GSL's
//           gsl_multimin_fdfminimizer_conjugate_fr (f-only trial points, df at accepted points), which is what the
//           reference drives (src/frontend/local_optim_contrast_gsl.cpp:80-215, src/backend/global_optim_contrast_gsl.cpp:20-115);
//   device (cmx_kernels.hip, finalize_body): the finalize step of an evaluation feeds it the cost / gradient it has just
//           reduced and writes the NEXT evaluation point to device memory, so that the next evaluation's kernels -- queued
//           by the host ahead of time -- start without a round trip to the host (cmx_chain.cpp, cmx_frontend_solve).
// The algorithm is a restatement of GSL's multimin/conjugate_fr.c + directional_minimize.c (take_step, intermediate_point,
// minimize) and of the stopping rules of the reference's loops.  GSL is un-vendored and absent here: PARITY UNPINNED against
// GSL itself; a second, independently written restatement (oracle/frcg.py) is run against this one call for call in
// tests/test_frcg_independent.py.  By-value parameters of the GSL routines that are modified inside them (fc / stepc inside
// intermediate_point; stepa, stepb, stepc, fa, fb, fc inside minimize) are separate fields here, as they are separate
// variables there.
#pragma once
#include <math.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define CMX_SM_HD __host__ __device__
#else
#define CMX_SM_HD
#endif

namespace cmx {

enum { FRCG_SUCCESS = 0, FRCG_CONTINUE = -2, FRCG_ENOPROG = 27 };  // GSL_SUCCESS / GSL_CONTINUE / GSL_ENOPROG

// what the machine is waiting for
enum SmPhase {
  SM_INIT = 0,     // fdf at x0 (gsl_multimin_fdfminimizer_set)
  SM_TRIAL = 1,    // f at the trial point of iterate(); df at the same point iff fc < fa
  SM_TRIAL_G = 2,
  SM_IP = 3,       // f inside intermediate_point; df iff !(fb >= fa && stepb > 0)
  SM_IP_G = 4,
  SM_IP_EQ_G = 5,  // intermediate_point's "trial point did not move": df only
  SM_MIN = 6,      // f inside minimize; df iff fm <= fb
  SM_MIN_G = 7,
  SM_DONE = 8
};
// kinds of request
enum { SM_REQ_F_GATED = 0, SM_REQ_FDF = 1, SM_REQ_DF = 2 };

// everything but the vectors: shared by the two storage forms below
struct FrcgScalars {
  // configuration
  int max_iter, dir_iter;
  double step_size, tol, epsabs_grad, tolfun;
  // minimiser state (conjugate_fr_state_t)
  double f, step, max_step, pnorm, g0norm, g1norm;
  // iterate()'s locals that live across evaluations
  double fa, fb, fc, dir, stepa, stepb, stepc, pg;
  // intermediate_point()'s by-value copies
  double ip_fc, ip_stepc;
  // minimize()'s by-value copies and locals
  double m_stepa, m_stepb, m_stepc, m_fa, m_fb, m_fc, u, v, w, fu, fv, fw, old1, old2, stepm, fm;
  // the reference's driver loop
  double cost_new, cost_old, initial_cost;
  int m_iter, iter, status, n_f, n_df;
  // current request
  int phase, req_kind, gate_mode;  // gate: the test on the cost f that decides whether df follows (1: f < thr, 2: f <= thr, 3: !(f >= thr), 4: always)
  int req_at_x;                    // 1: the evaluation point is x (SM_INIT), 0: x1
  int pad_;
  double gate_thr;
};
// host form: any n, the 9 x n vectors owned by the caller (one contiguous block x | gradient | dx | x1 | dx1 | x2 | dx2 | p | g0)
struct FrcgSM : FrcgScalars {
  int n;
  double *x, *gradient, *dx, *x1, *dx1, *x2, *dx2, *p, *g0;
};
// device form: n known at compile time, vectors inside the struct -- one block of memory, every access a fixed offset (the
// finalize step keeps it in LDS while it advances it), every loop unrolled
template <int N>
struct FrcgSMFix : FrcgScalars {
  static constexpr int n = N;
  double x[N], gradient[N], dx[N], x1[N], dx1[N], x2[N], dx2[N], p[N], g0[N];
};

template <class SM>
CMX_SM_HD inline const double *sm_point(const SM &s) { return s.req_at_x ? s.x : s.x1; }
template <class SM>
CMX_SM_HD inline bool sm_done(const SM &s) { return s.phase == SM_DONE; }

// gsl_blas_dnrm2 as GSL's own CBLAS computes it (cblas/source_nrm2_r.h): running scale + scaled sum of squares
CMX_SM_HD inline double sm_nrm2(const double *v, int n) {
  if (n <= 0) return 0.0;
  if (n == 1) return fabs(v[0]);
  double scale = 0.0, ssq = 1.0;
  for (int i = 0; i < n; i++) {
    const double e = v[i];
    if (e != 0.0) {
      const double ax = fabs(e);
      if (scale < ax) {
        ssq = 1.0 + ssq * (scale / ax) * (scale / ax);
        scale = ax;
      } else {
        ssq += (ax / scale) * (ax / scale);
      }
    }
  }
  return scale * sqrt(ssq);
}
CMX_SM_HD inline double sm_dot(const double *a, const double *b, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
CMX_SM_HD inline void sm_copy(double *d, const double *s, int n) {
  for (int i = 0; i < n; i++) d[i] = s[i];
}
// x1 = x - step*lambda*p ; dx = -step*lambda*p   (take_step: gsl_vector_set_zero, daxpy, memcpy, daxpy)
template <class SM>
CMX_SM_HD inline void sm_take_step(const SM &s, double step, double lambda, double *xo, double *dxo) {
  for (int i = 0; i < s.n; i++) dxo[i] = 0.0;
  for (int i = 0; i < s.n; i++) dxo[i] += -step * lambda * s.p[i];
  for (int i = 0; i < s.n; i++) xo[i] = s.x[i] + 1.0 * dxo[i];
}
CMX_SM_HD inline double sm_sq(double r) {
#if defined(__HIP_DEVICE_COMPILE__)
  return r * r;  // (no glibc on the device: differs from pow(r, 2.0) in the last bit for ~0.08 % of arguments; the host replays
                 //  every step with pow and takes over when the device's next point is not bitwise its own, cmx_chain.cpp)
#else
  return pow(r, 2.0);  // conjugate_fr.c: double beta = -pow (g1norm / g0norm, 2.0);
#endif
}

template <class SM>
CMX_SM_HD inline void sm_request(SM &s, int phase, int kind, int mode, double thr) {
  s.phase = phase;
  s.req_kind = kind;
  s.gate_mode = mode;
  s.gate_thr = thr;
  s.req_at_x = 0;
}

template <class SM>
CMX_SM_HD inline void sm_finish(SM &s, int status) {
  s.status = status;
  s.phase = SM_DONE;
}

// ---- minimize(): one pass of its loop up to the next evaluation, or its return
template <class SM>
CMX_SM_HD inline void sm_driver_post(SM &s, int status);
template <class SM>
CMX_SM_HD inline void sm_post_minimize(SM &s) {
  // back in iterate(): x = x2, new conjugate direction
  sm_copy(s.x, s.x2, s.n);
  s.dir_iter = (s.dir_iter + 1) % s.n;
  if (s.dir_iter == 0) {
    sm_copy(s.p, s.gradient, s.n);
    s.pnorm = s.g1norm;
  } else {
    const double beta = -sm_sq(s.g1norm / s.g0norm);  // p' = g1 - beta * p
    for (int i = 0; i < s.n; i++) s.p[i] = -beta * s.p[i];
    for (int i = 0; i < s.n; i++) s.p[i] += s.gradient[i];
    s.pnorm = sm_nrm2(s.p, s.n);
  }
  s.g0norm = s.g1norm;
  sm_copy(s.g0, s.gradient, s.n);
  sm_driver_post(s, FRCG_SUCCESS);
}
template <class SM>
CMX_SM_HD inline void sm_min_next(SM &s) {
  s.m_iter++;
  if (s.m_iter > 10) {  // MAX ITERATIONS
    sm_post_minimize(s);
    return;
  }
  const double dw = s.w - s.u, dv = s.v - s.u;
  double du = 0.0;
  const double e1 = ((s.fv - s.fu) * dw * dw + (s.fu - s.fw) * dv * dv);
  const double e2 = 2.0 * ((s.fv - s.fu) * dw + (s.fu - s.fw) * dv);
  if (e2 != 0.0) du = e1 / e2;
  if (du > 0.0 && du < (s.m_stepc - s.m_stepb) && fabs(du) < 0.5 * s.old2) s.stepm = s.u + du;
  else if (du < 0.0 && du > (s.m_stepa - s.m_stepb) && fabs(du) < 0.5 * s.old2) s.stepm = s.u + du;
  else if ((s.m_stepc - s.m_stepb) > (s.m_stepb - s.m_stepa)) s.stepm = 0.38 * (s.m_stepc - s.m_stepb) + s.m_stepb;
  else s.stepm = s.m_stepb - 0.38 * (s.m_stepb - s.m_stepa);
  sm_take_step(s, s.stepm, s.dir / s.pnorm, s.x1, s.dx1);
  sm_request(s, SM_MIN, SM_REQ_F_GATED, 2, s.m_fb);  // df(x1) follows iff fm <= fb
}
template <class SM>
CMX_SM_HD inline void sm_min_begin(SM &s) {
  // minimize (p, x, dir / pnorm, stepa, stepb, stepc, fa, fb, fc, tol, x1, dx1, x2, dx, gradient, &step, &f, &g1norm)
  s.m_stepa = s.stepa; s.m_stepb = s.stepb; s.m_stepc = s.stepc;
  s.m_fa = s.fa; s.m_fb = s.fb; s.m_fc = s.fc;
  s.u = s.m_stepb; s.v = s.m_stepa; s.w = s.m_stepc;
  s.fu = s.m_fb; s.fv = s.m_fa; s.fw = s.m_fc;
  s.old2 = fabs(s.w - s.v);
  s.old1 = fabs(s.v - s.u);
  s.m_iter = 0;
  sm_copy(s.x2, s.x1, s.n);
  // DELIBERATE DEVIATION in a value nothing reads: in GSL the callee's `dx2` parameter IS the state's `dx` (iterate() passes it
  // for both), so GSL's memcpy(dx2, dx1) leaves dx = dx1 when no step of minimize() is accepted; here dx2 is its own field and
  // dx keeps the trial step's value.  m_fa / m_fc are likewise not refreshed on rejected steps.  The reference's stopping rules
  // (local_optim_contrast_gsl.cpp:134-215) read the gradient and f only, never dx: a host that wants GSL's dx after an
  // unsuccessful line search must not take it from this machine.
  sm_copy(s.dx2, s.dx, s.n);
  s.f = s.m_fb;
  s.step = s.m_stepb;
  s.g1norm = sm_nrm2(s.gradient, s.n);
  sm_min_next(s);
}

// ---- intermediate_point(): one pass of its loop up to the next evaluation
template <class SM>
CMX_SM_HD inline void sm_ip_next(SM &s) {
  const double lambda = s.dir / s.pnorm;
  const double u = fabs(s.pg * lambda * s.ip_stepc);
  s.stepb = 0.5 * s.ip_stepc * u / ((s.ip_fc - s.fa) + u);
  sm_take_step(s, s.stepb, lambda, s.x1, s.dx);
  bool equal = true;
  for (int i = 0; i < s.n; i++)
    if (s.x[i] != s.x1[i]) { equal = false; break; }
  if (equal) {  // trial point did not move from the initial point: *step = 0, *f = fa, df(x1)
    sm_request(s, SM_IP_EQ_G, SM_REQ_DF, 4, 0.0);
    return;
  }
  sm_request(s, SM_IP, SM_REQ_F_GATED, s.stepb > 0.0 ? 3 : 4, s.fa);  // df(x1) follows unless (fb >= fa && stepb > 0)
}

// ---- iterate(): from its entry to its first evaluation
template <class SM>
CMX_SM_HD inline void sm_iterate_begin(SM &s) {
  s.fa = s.f;
  s.stepa = 0.0;
  s.stepc = s.step;
  if (s.pnorm == 0.0 || s.g0norm == 0.0) {
    for (int i = 0; i < s.n; i++) s.dx[i] = 0.0;
    sm_driver_post(s, FRCG_ENOPROG);
    return;
  }
  s.pg = sm_dot(s.p, s.gradient, s.n);  // which direction is downhill, +p or -p
  s.dir = (s.pg >= 0.0) ? +1.0 : -1.0;
  sm_take_step(s, s.stepc, s.dir / s.pnorm, s.x1, s.dx);  // trial point x_c = x - step * p
  sm_request(s, SM_TRIAL, SM_REQ_F_GATED, 1, s.fa);       // df(x1) follows iff fc < fa
}

// ---- the reference's loop around gsl_multimin_fdfminimizer_iterate (local_optim_contrast_gsl.cpp:134-215)
template <class SM>
CMX_SM_HD inline void sm_driver_next(SM &s) {
  s.iter++;
  s.cost_old = s.cost_new;
  sm_iterate_begin(s);
}
template <class SM>
CMX_SM_HD inline void sm_driver_post(SM &s, int status) {
  if (status == FRCG_SUCCESS) {
    s.cost_new = s.f;  // convergence due to stagnation in the value of the function
    if (fabs(1 - s.cost_new / (s.cost_old + 1e-7)) < s.tolfun) { sm_finish(s, status); return; }
    status = FRCG_CONTINUE;
  }
  if (sm_nrm2(s.gradient, s.n) < s.epsabs_grad) { sm_finish(s, status); return; }  // gsl_multimin_test_gradient
  if (status != FRCG_CONTINUE) { sm_finish(s, status); return; }                   // the iteration did not reduce the value
  if (s.iter < s.max_iter) sm_driver_next(s);
  else sm_finish(s, status);
}

// ---- entry points -------------------------------------------------------------------------------------------------
// vectors must be set (9 x n doubles), x holds the start point
CMX_SM_HD inline void sm_begin(FrcgSM &s, int n, double step_size, double tol, double epsabs_grad, double tolfun, int max_iter) {
  s.n = n; s.max_iter = max_iter;
  s.step_size = step_size; s.tol = tol; s.epsabs_grad = epsabs_grad; s.tolfun = tolfun;
  for (int i = 0; i < n; i++) { s.gradient[i] = 0; s.dx[i] = 0; s.x1[i] = 0; s.dx1[i] = 0; s.x2[i] = 0; s.dx2[i] = 0; s.p[i] = 0; s.g0[i] = 0; }
  s.dir_iter = 0;
  s.step = step_size;
  s.max_step = step_size;
  s.f = 0; s.pnorm = 0; s.g0norm = 0; s.g1norm = 0;
  s.fa = s.fb = s.fc = s.dir = s.stepa = s.stepb = s.stepc = s.pg = s.ip_fc = s.ip_stepc = 0;
  s.m_stepa = s.m_stepb = s.m_stepc = s.m_fa = s.m_fb = s.m_fc = s.u = s.v = s.w = s.fu = s.fv = s.fw = s.old1 = s.old2 = s.stepm = s.fm = 0;
  s.m_iter = 0;
  s.iter = 0; s.status = FRCG_CONTINUE; s.n_f = 0; s.n_df = 0;
  s.cost_new = 1e9; s.cost_old = 1e9; s.initial_cost = 0;
  s.phase = SM_INIT; s.req_kind = SM_REQ_FDF; s.gate_mode = 4; s.gate_thr = 0; s.req_at_x = 1;
}

// The cost f at the requested point (ignored for SM_REQ_DF).  Returns true iff the gradient at the same point is needed next
// (then call sm_grad); otherwise the machine has already moved on to the next request (or finished).
template <class SM>
CMX_SM_HD inline bool sm_cost(SM &s, double fval) {
  switch (s.phase) {
    case SM_INIT:
      s.f = fval;
      return true;
    case SM_TRIAL:
      s.n_f++;
      s.fc = fval;
      if (s.fc < s.fa) { s.phase = SM_TRIAL_G; return true; }  // success: reduced the function value
      // line minimisation in (xa,fa) (xc,fc): find an intermediate (xb,fb) with fa > fb < fc
      s.ip_fc = s.fc;
      s.ip_stepc = s.stepc;
      sm_ip_next(s);
      return false;  // (the next request may be intermediate_point's df-only at a point that did not move: a NEW point)
    case SM_IP:
      s.n_f++;
      s.fb = fval;
      if (s.fb >= s.fa && s.stepb > 0.0) {  // downhill step failed: reduce the step and try again
        s.ip_fc = s.fb;
        s.ip_stepc = s.stepb;
        sm_ip_next(s);
        return false;
      }
      s.phase = SM_IP_G;
      return true;
    case SM_IP_EQ_G:
      return true;
    case SM_MIN:
      s.n_f++;
      s.fm = fval;
      if (s.fm > s.m_fb) {
        if (s.fm < s.fv) { s.w = s.v; s.v = s.stepm; s.fw = s.fv; s.fv = s.fm; }
        else if (s.fm < s.fw) { s.w = s.stepm; s.fw = s.fm; }
        if (s.stepm < s.m_stepb) s.m_stepa = s.stepm;
        else s.m_stepc = s.stepm;
        sm_min_next(s);
        return false;
      } else if (s.fm <= s.m_fb) {
        s.phase = SM_MIN_G;
        return true;
      }
      sm_post_minimize(s);  // fm is NaN (a failed evaluation): GSL's if / else-if pair takes neither branch and falls out
      return false;
    default:
      return false;
  }
}

// The gradient at the point whose cost was just fed (sm_cost returned true)
template <class SM>
CMX_SM_HD inline void sm_grad(SM &s, const double *g) {
  s.n_df++;
  switch (s.phase) {
    case SM_INIT: {  // gsl_multimin_fdfminimizer_set: first direction = gradient
      sm_copy(s.gradient, g, s.n);
      sm_copy(s.p, g, s.n);
      sm_copy(s.g0, g, s.n);
      const double gnorm = sm_nrm2(s.gradient, s.n);
      s.pnorm = gnorm;
      s.g0norm = gnorm;
      s.initial_cost = s.f;
      s.iter = 0;
      s.cost_new = 1e9;
      s.cost_old = 1e9;
      s.status = FRCG_CONTINUE;
      sm_driver_next(s);
      return;
    }
    case SM_TRIAL_G:
      s.step = s.stepc * 2.0;
      s.f = s.fc;
      sm_copy(s.x, s.x1, s.n);
      sm_copy(s.gradient, g, s.n);
      sm_driver_post(s, FRCG_SUCCESS);
      return;
    case SM_IP_EQ_G:
      sm_copy(s.gradient, g, s.n);
      s.stepb = 0;
      s.fb = s.fa;
      sm_driver_post(s, FRCG_ENOPROG);  // if (stepb == 0.0) return GSL_ENOPROG;
      return;
    case SM_IP_G:
      sm_copy(s.gradient, g, s.n);
      if (s.stepb == 0.0) { sm_driver_post(s, FRCG_ENOPROG); return; }
      sm_min_begin(s);
      return;
    case SM_MIN_G: {
      s.old2 = s.old1;
      s.old1 = fabs(s.u - s.stepm);
      s.w = s.v; s.v = s.u; s.u = s.stepm;
      s.fw = s.fv; s.fv = s.fu; s.fu = s.fm;
      sm_copy(s.x2, s.x1, s.n);
      sm_copy(s.dx2, s.dx1, s.n);
      sm_copy(s.gradient, g, s.n);
      const double pg = sm_dot(s.p, s.gradient, s.n);
      const double gnorm1 = sm_nrm2(s.gradient, s.n);
      s.f = s.fm;
      s.step = s.stepm;
      s.g1norm = gnorm1;
      sm_copy(s.dx, s.dx2, s.n);
      if (fabs(pg * (s.dir / s.pnorm) / gnorm1) < s.tol) { sm_post_minimize(s); return; }  // SUCCESS
      if (s.stepm < s.m_stepb) { s.m_stepc = s.m_stepb; s.m_fc = s.m_fb; s.m_stepb = s.stepm; s.m_fb = s.fm; }
      else { s.m_stepa = s.m_stepb; s.m_fa = s.m_fb; s.m_stepb = s.stepm; s.m_fb = s.fm; }
      sm_min_next(s);
      return;
    }
    default:
      return;
  }
}

// the device form of a host machine (front end: N = 3)
template <int N>
inline void sm_to_fixed(const FrcgSM &h, FrcgSMFix<N> &d) {
  static_cast<FrcgScalars &>(d) = static_cast<const FrcgScalars &>(h);
  for (int i = 0; i < N; i++) {
    d.x[i] = h.x[i]; d.gradient[i] = h.gradient[i]; d.dx[i] = h.dx[i]; d.x1[i] = h.x1[i]; d.dx1[i] = h.dx1[i];
    d.x2[i] = h.x2[i]; d.dx2[i] = h.dx2[i]; d.p[i] = h.p[i]; d.g0[i] = h.g0[i];
  }
}

// same shape as gsl_multimin_function_fdf
struct FunctionFdf {
  double (*f)(const double *x, void *params);
  void (*df)(const double *x, void *params, double *g);
  void (*fdf)(const double *x, void *params, double *f, double *g);
  size_t n;
  void *params;
  // optional (not in GSL): called right before a cost-only evaluation with the test on its value f that decides whether the
  // gradient at the same point is requested next -- mode 1: f < thr, 2: f <= thr, 3: !(f >= thr), 4: always.  An evaluator
  // that can act on it (cmx_hint_next_df) queues the gradient pass behind the cost evaluation; the sequence of f / df
  // calls is unchanged.
  void (*hint)(double thr, int mode, void *params) = nullptr;
};

// One step of the host-driven form: performs the machine's current request through the callbacks -- exactly the calls GSL's
// conjugate_fr makes (f at trial points, df at accepted points, fdf at the start) -- and feeds the results back.
inline void sm_step_host(FrcgSM &s, const FunctionFdf &fn, double *g_scratch) {
  const double *xr = sm_point(s);
  if (s.req_kind == SM_REQ_FDF) {
    double fval = 0;
    fn.fdf(xr, fn.params, &fval, g_scratch);
    (void)sm_cost(s, fval);
    sm_grad(s, g_scratch);
  } else if (s.req_kind == SM_REQ_DF) {
    fn.df(xr, fn.params, g_scratch);
    (void)sm_cost(s, 0.0);
    sm_grad(s, g_scratch);
  } else {
    if (fn.hint) fn.hint(s.gate_thr, s.gate_mode, fn.params);
    const double fval = fn.f(xr, fn.params);
    if (sm_cost(s, fval)) {  // (x1 is unchanged when the gradient at the same point is requested)
      fn.df(sm_point(s), fn.params, g_scratch);
      sm_grad(s, g_scratch);
    }
  }
}

}  // namespace cmx
