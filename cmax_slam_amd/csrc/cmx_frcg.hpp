// cmx_frcg.hpp -- Fletcher-Reeves conjugate-gradient minimiser, host C++.
//
// The reference drives both cost functors with GSL's gsl_multimin_fdfminimizer_conjugate_fr
//   (src/frontend/local_optim_contrast_gsl.cpp:80,113,138; src/backend/global_optim_contrast_gsl.cpp:20,48,73).
// GSL is an un-vendored dependency (absent here): this is a restatement of the published algorithm of
// GSL's multimin/conjugate_fr.c + directional_minimize.c (take_step / intermediate_point / minimize),
// keeping its call pattern (f-only trial points, df at accepted points) because that pattern is what the
// evaluator is tuned for.  PARITY UNPINNED against GSL itself; the stopping rules of the reference's driver
// loops are restated in cmx_solver.cpp.  A second, independently written restatement (oracle/frcg.py, Python) is run
// against this one call for call in tests/test_frcg_independent.py.
#pragma once
#include <math.h>
#include <stddef.h>

#include <vector>

namespace cmx {

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

enum { FRCG_SUCCESS = 0, FRCG_CONTINUE = -2, FRCG_ENOPROG = 27 };  // GSL_SUCCESS / GSL_CONTINUE / GSL_ENOPROG

class FrcgMinimizer {
 public:
  // gsl_multimin_fdfminimizer_set: evaluates fdf at x0, first direction = gradient
  void set(const FunctionFdf &fn, const double *x0, double step_size, double tol) {
    fn_ = fn;
    n_ = fn.n;
    x.assign(x0, x0 + n_);
    gradient.assign(n_, 0.0);
    dx.assign(n_, 0.0);
    x1.assign(n_, 0.0); dx1.assign(n_, 0.0); x2.assign(n_, 0.0);
    p.assign(n_, 0.0); g0.assign(n_, 0.0);
    iter_ = 0;
    step_ = step_size;
    max_step_ = step_size;
    tol_ = tol;
    fn_.fdf(x.data(), fn_.params, &f, gradient.data());
    p = gradient;
    g0 = gradient;
    const double gnorm = nrm2(gradient);
    pnorm_ = gnorm;
    g0norm_ = gnorm;
  }

  // gsl_multimin_fdfminimizer_iterate (conjugate_fr_iterate)
  int iterate() {
    double fa = f, fb, fc;
    double dir;
    double stepa = 0.0, stepb, stepc = step_, tol = tol_;
    double g1norm;
    double pg;
    if (pnorm_ == 0.0 || g0norm_ == 0.0) {
      for (auto &v : dx) v = 0;
      return FRCG_ENOPROG;
    }
    // which direction is downhill, +p or -p
    pg = dot(p, gradient);
    dir = (pg >= 0.0) ? +1.0 : -1.0;
    // trial point x_c = x - step * p
    take_step(x, p, stepc, dir / pnorm_, x1, dx);
    if (fn_.hint) fn_.hint(fa, 1, fn_.params);  // df(x1) follows iff fc < fa
    fc = fn_.f(x1.data(), fn_.params);
    if (fc < fa) {
      // success: reduced the function value
      step_ = stepc * 2.0;
      f = fc;
      x = x1;
      fn_.df(x1.data(), fn_.params, gradient.data());
      return FRCG_SUCCESS;
    }
    // line minimisation in (xa,fa) (xc,fc): find an intermediate (xb,fb) with fa > fb < fc
    intermediate_point(dir / pnorm_, pg, stepa, stepc, fa, fc, &stepb, &fb);
    if (stepb == 0.0) return FRCG_ENOPROG;
    minimize(dir / pnorm_, stepa, stepb, stepc, fa, fb, fc, tol, &step_, &f, &g1norm);
    x = x2;
    // new conjugate direction
    iter_ = (iter_ + 1) % n_;
    if (iter_ == 0) {
      p = gradient;
      pnorm_ = g1norm;
    } else {
      // p' = g1 - beta * p
      const double beta = -pow(g1norm / g0norm_, 2.0);
      for (size_t i = 0; i < n_; i++) p[i] = -beta * p[i];
      for (size_t i = 0; i < n_; i++) p[i] += gradient[i];
      pnorm_ = nrm2(p);
    }
    g0norm_ = g1norm;
    g0 = gradient;
    return FRCG_SUCCESS;
  }

  // gsl_blas_dnrm2 as GSL's own CBLAS computes it (cblas/source_nrm2_r.h): running scale + scaled sum of squares
  static double nrm2(const std::vector<double> &v) {
    if (v.empty()) return 0.0;
    if (v.size() == 1) return fabs(v[0]);
    double scale = 0.0, ssq = 1.0;
    for (double e : v) {
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

  std::vector<double> x, gradient, dx;
  double f = 0;

 private:
  static double dot(const std::vector<double> &a, const std::vector<double> &b) {
    double s = 0;
    for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
    return s;
  }
  // x1 = x - step*lambda*p ; dx = -step*lambda*p
  void take_step(const std::vector<double> &x0, const std::vector<double> &pp, double step, double lambda,
                 std::vector<double> &xo, std::vector<double> &dxo) const {
    for (size_t i = 0; i < n_; i++) dxo[i] = 0.0;
    for (size_t i = 0; i < n_; i++) dxo[i] += -step * lambda * pp[i];
    for (size_t i = 0; i < n_; i++) xo[i] = x0[i] + 1.0 * dxo[i];
  }

  void intermediate_point(double lambda, double pg, double stepa, double stepc, double fa, double fc, double *step,
                          double *fout) {
    double stepb, fb;
    (void)stepa;
    for (;;) {
      const double u = fabs(pg * lambda * stepc);
      stepb = 0.5 * stepc * u / ((fc - fa) + u);
      take_step(x, p, stepb, lambda, x1, dx);
      bool equal = true;
      for (size_t i = 0; i < n_; i++)
        if (x[i] != x1[i]) { equal = false; break; }
      if (equal) {
        // trial point did not move from the initial point
        *step = 0;
        *fout = fa;
        fn_.df(x1.data(), fn_.params, gradient.data());
        return;
      }
      if (fn_.hint) fn_.hint(fa, stepb > 0.0 ? 3 : 4, fn_.params);  // df(x1) follows unless (fb >= fa && stepb > 0)
      fb = fn_.f(x1.data(), fn_.params);
      if (fb >= fa && stepb > 0.0) {
        // downhill step failed: reduce the step and try again
        fc = fb;
        stepc = stepb;
        continue;
      }
      break;
    }
    *step = stepb;
    *fout = fb;
    fn_.df(x1.data(), fn_.params, gradient.data());
  }

  void minimize(double lambda, double stepa, double stepb, double stepc, double fa, double fb, double fc, double tol,
                double *step, double *fout, double *gnorm) {
    // starting at (x, f) move along p to find a minimum f(x - lambda*step*p); Brent-like with parabolic steps
    double u = stepb, v = stepa, w = stepc;
    double fu = fb, fv = fa, fw = fc;
    double old2 = fabs(w - v);
    double old1 = fabs(v - u);
    double stepm, fm, pg, gnorm1;
    int iter = 0;
    x2 = x1;
    std::vector<double> dx2 = dx;
    *fout = fb;
    *step = stepb;
    *gnorm = nrm2(gradient);
    for (;;) {
      iter++;
      if (iter > 10) return;  // MAX ITERATIONS
      {
        const double dw = w - u, dv = v - u;
        double du = 0.0;
        const double e1 = ((fv - fu) * dw * dw + (fu - fw) * dv * dv);
        const double e2 = 2.0 * ((fv - fu) * dw + (fu - fw) * dv);
        if (e2 != 0.0) du = e1 / e2;
        if (du > 0.0 && du < (stepc - stepb) && fabs(du) < 0.5 * old2) stepm = u + du;
        else if (du < 0.0 && du > (stepa - stepb) && fabs(du) < 0.5 * old2) stepm = u + du;
        else if ((stepc - stepb) > (stepb - stepa)) stepm = 0.38 * (stepc - stepb) + stepb;
        else stepm = stepb - 0.38 * (stepb - stepa);
      }
      take_step(x, p, stepm, lambda, x1, dx1);
      if (fn_.hint) fn_.hint(fb, 2, fn_.params);  // df(x1) follows iff fm <= fb
      fm = fn_.f(x1.data(), fn_.params);
      if (fm > fb) {
        if (fm < fv) { w = v; v = stepm; fw = fv; fv = fm; }
        else if (fm < fw) { w = stepm; fw = fm; }
        if (stepm < stepb) stepa = stepm;
        else stepc = stepm;
        continue;
      } else if (fm <= fb) {
        old2 = old1;
        old1 = fabs(u - stepm);
        w = v; v = u; u = stepm;
        fw = fv; fv = fu; fu = fm;
        x2 = x1;
        dx2 = dx1;
        fn_.df(x1.data(), fn_.params, gradient.data());
        pg = dot(p, gradient);
        gnorm1 = nrm2(gradient);
        *fout = fm;
        *step = stepm;
        *gnorm = gnorm1;
        dx = dx2;
        if (fabs(pg * lambda / gnorm1) < tol) return;  // SUCCESS
        if (stepm < stepb) { stepc = stepb; fc = fb; stepb = stepm; fb = fm; }
        else { stepa = stepb; fa = fb; stepb = stepm; fb = fm; }
        continue;
      } else {
        return;  // fm is NaN (a failed evaluation): GSL's if / else-if pair takes neither branch and falls out of the function
      }
    }
  }

  FunctionFdf fn_{};
  size_t n_ = 0;
  int iter_ = 0;
  double step_ = 0, max_step_ = 0, tol_ = 0, pnorm_ = 0, g0norm_ = 0;
  std::vector<double> x1, dx1, x2, p, g0;
};

}  // namespace cmx
