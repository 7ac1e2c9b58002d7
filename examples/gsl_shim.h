/* gsl_shim.h -- layout-compatible stand-ins for the three GSL types the reference's cost functors are written against
 * (gsl_vector, gsl_block, gsl_multimin_function_fdf; GSL's public <gsl/gsl_vector_double.h> / <gsl/gsl_multimin.h>), for hosts
 * built where GSL is not installed -- this image, the GPU box.  With GSL present include its headers instead: this file then
 * defines nothing (the include guards below are GSL's own), and the callbacks of examples/gsl_style_host.cpp compile unchanged
 * against the real types -- they carry the reference's exact signatures (src/frontend/local_optim_contrast_gsl.cpp:19-70,
 * src/backend/global_optim_contrast_gsl_analytical.cpp:17-81):
 *     void   fdf(const gsl_vector *v, void *params, double *f, gsl_vector *df);
 *     double f  (const gsl_vector *v, void *params);
 *     void   df (const gsl_vector *v, void *params, gsl_vector *df);
 * Only what those bodies touch is here: the struct layouts, gsl_vector_get / gsl_vector_set, a stack view constructor. */
#ifndef CMX_EXAMPLES_GSL_SHIM_H
#define CMX_EXAMPLES_GSL_SHIM_H
#include <stddef.h>

#ifndef __GSL_VECTOR_DOUBLE_H__
#define __GSL_VECTOR_DOUBLE_H__
typedef struct {
  size_t size;
  double *data;
} gsl_block;
typedef struct {
  size_t size;
  size_t stride;
  double *data;
  gsl_block *block;
  int owner;
} gsl_vector;
static inline double gsl_vector_get(const gsl_vector *v, const size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_set(gsl_vector *v, const size_t i, double x) { v->data[i * v->stride] = x; }
#endif

#ifndef __GSL_MULTIMIN_H__
#define __GSL_MULTIMIN_H__
typedef struct {
  double (*f)(const gsl_vector *x, void *params);
  void (*df)(const gsl_vector *x, void *params, gsl_vector *df);
  void (*fdf)(const gsl_vector *x, void *params, double *f, gsl_vector *df);
  size_t n;
  void *params;
} gsl_multimin_function_fdf;
#endif

/* a gsl_vector over caller-owned doubles (what gsl_vector_view_array produces), for adapters between the pointer-based driver
 * of this library (cmx_frcg_minimize*) and functors written against gsl_vector */
static inline gsl_vector cmx_gsl_view(double *data, size_t n) {
  gsl_vector v;
  v.size = n;
  v.stride = 1;
  v.data = data;
  v.block = 0;
  v.owner = 0;
  return v;
}
#endif /* CMX_EXAMPLES_GSL_SHIM_H */
