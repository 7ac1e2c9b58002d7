/*
 * exact_f64.c -- the oracle's own sources with every `float` turned into `double`: the reference's algorithm without
 * fp32 rounding (images, bilinear weights, Jacobian chain, blur taps and accumulators all fp64).
 *
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).  Used as the arbiter where the fp32 oracle -- i.e. the reference's own
 * arithmetic -- is itself further than north_star's 1e-5 from the value its formula defines: near a stationary point
 * the analytic gradient is a small difference of large fp32 sums, and over 250 random back-end configurations the
 * fp32 oracle is up to 3.0e-5 of |g|_inf away from this build (tests/exact_noise.py).
 *
 * One translation unit: the system headers come first so that their prototypes keep `float`; the `f`-suffixed libm
 * calls the sources make on image values are mapped to their double forms.  The exported functions have the ABI of
 * cmax_oracle.h with `float *` read as `double *`.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define float double
#define expf exp
#define sqrtf sqrt
#define fabsf fabs
#define floorf floor

#include "cv_ops.c"
#include "frontend.c"
#include "backend.c"
#include "so3_spline.c"
#include "traj_init.c"
