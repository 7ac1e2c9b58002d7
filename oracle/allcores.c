/*
 * allcores.c -- OpenMP all-cores variant of the CPU restatement.  NOT THE REFERENCE: cmax_slam runs each path on
 * one thread (src/node.cpp:22, src/cmax_slam.cpp:92; no `#pragma omp` anywhere in the reference).  This file exists
 * only so that the benchmark can print "what the same algorithm does on every host core" beside the faithful
 * single-thread number (CPU baseline (ii)).
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).  Built into liboracle_mt.so with -fopenmp.
 *
 * Method: contiguous ranges of whole event batches per thread (so every batch keeps the pose the reference gives
 * it), thread-private fp32 images, summed in thread order; the image passes (blur rows, per-plane blurs and
 * gradient sums) are split over threads without changing any element's arithmetic; only the fp64 image sums use an
 * OpenMP reduction (different summation order: 1e-16 relative).  Results therefore differ from the
 * single-thread oracle only through the fp32 vote order, like the GPU path.
 */
#include "cmax_oracle.h"
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* scratch kept between calls (a benchmark loop must not time page faults) */
static float *g_ws = NULL;
static size_t g_ws_len = 0;
static float *workspace(size_t len) {
  if (len > g_ws_len) {
    free(g_ws);
    g_ws = (float *)malloc(len * sizeof(float));
    g_ws_len = g_ws ? len : 0;
  }
  return g_ws;
}

int orc_mt_max_threads(void) { return omp_get_max_threads(); }

static int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

/* same element arithmetic as orc_gaussian_blur (cv_ops.c), rows split over threads */
static void blur_rows_mt(float *img, float *tmp, int W, int H, int cn, double sigma, int T) {
  const int n = orc_gauss_ksize(sigma), r = n / 2;
  float kx[64];
  if (n > 64) return;
  orc_gauss_kernel(n, sigma, kx);
  const size_t rowlen = (size_t)W * cn;
  int *xi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * r));
  for (int i = 0; i < W + 2 * r; i++) xi[i] = reflect101(i - r, W);
#pragma omp parallel for num_threads(T) schedule(static)
  for (int y = 0; y < H; y++) {
    const float *S = img + (size_t)y * rowlen;
    float *D = tmp + (size_t)y * rowlen;
    for (int x = 0; x < W; x++)
      for (int c = 0; c < cn; c++) {
        float s0 = kx[0] * S[(size_t)xi[x] * cn + c];
        for (int k = 1; k < n; k++) s0 += kx[k] * S[(size_t)xi[x + k] * cn + c];
        D[(size_t)x * cn + c] = s0;
      }
  }
  const float *ky = kx + r;
#pragma omp parallel for num_threads(T) schedule(static)
  for (int y = 0; y < H; y++) {
    float *D = img + (size_t)y * rowlen;
    const float *Sc = tmp + (size_t)y * rowlen;
    for (size_t i = 0; i < rowlen; i++) D[i] = ky[0] * Sc[i];
    for (int k = 1; k <= r; k++) {
      const float *Sp = tmp + (size_t)reflect101(y + k, H) * rowlen;
      const float *Sm = tmp + (size_t)reflect101(y - k, H) * rowlen;
      const float f = ky[k];
      for (size_t i = 0; i < rowlen; i++) D[i] += f * (Sp[i] + Sm[i]);
    }
  }
  free(xi);
}

/* contrast_Variance / contrast_MeanSquare (cv_ops.c) with the pixel sums as OpenMP reductions */
static double contrast_mt(const float *img, size_t npix, const float *const *ch, size_t stride, int P, int measure,
                          double *grad, int T) {
  double s = 0, sq = 0;
#pragma omp parallel for num_threads(T) reduction(+ : s, sq) schedule(static)
  for (size_t i = 0; i < npix; i++) {
    const double v = img[i];
    s += v;
    sq += v * v;
  }
  if (measure == ORC_MEAN_SQUARE) {
    if (grad)
      for (int k = 0; k < P; k++) {
        double a = 0;
#pragma omp parallel for num_threads(T) reduction(+ : a) schedule(static)
        for (size_t i = 0; i < npix; i++) {
          const float m = img[i] * ch[k][i * stride];
          a += m;
        }
        grad[k] = 2. * (a / (double)npix);
      }
    return sq / (double)npix;
  }
  const double mean = s / (double)npix;
  double var = sq / (double)npix - mean * mean;
  if (var < 0) var = 0;
  const double sd = sqrt(var);
  if (grad) {
    const float beta = (float)(-2. * mean);
    for (int k = 0; k < P; k++) {
      double sk = 0;
#pragma omp parallel for num_threads(T) reduction(+ : sk) schedule(static)
      for (size_t i = 0; i < npix; i++) sk += ch[k][i * stride];
      const float mkf = (float)(sk / (double)npix);
      double acc = 0;
#pragma omp parallel for num_threads(T) reduction(+ : acc) schedule(static)
      for (size_t i = 0; i < npix; i++) {
        const float z = img[i] * 2.f + beta;
        const float d = ch[k][i * stride] - mkf;
        const float m = z * d;
        acc += m;
      }
      grad[k] = acc / (double)npix;
    }
  }
  return sd * sd;
}

/* front end: local_contrast_fdf on T threads */
int orc_fe_eval_mt(const orc_fe_cfg *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                   int64_t t_ref_ns, const double omega[3], int nthreads, double *contrast, double *grad) {
  const size_t np = (size_t)c->W * c->H;
  const size_t per = np * (grad ? 4 : 1); /* [iwe | deriv(3, interleaved)] per thread */
  if (c->measure == ORC_GRADIENT_MAGNITUDE) return -3;
  for (int64_t i = 0; i < n; i++)
    if (x[i] >= c->W || y[i] >= c->H) return -1;
  int T = nthreads > 0 ? nthreads : omp_get_max_threads();
  const int64_t nb = (n + c->batch - 1) / c->batch;
  if (T > nb) T = nb > 0 ? (int)nb : 1;
  float *ws = workspace(per * (size_t)(T + 1));
  if (!ws) return -4;
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    float *iwe = ws + per * (size_t)t, *deriv = grad ? iwe + np : NULL;
    memset(iwe, 0, per * sizeof(float));
    const int64_t b0 = nb * t / T, b1 = nb * (t + 1) / T;
    for (int64_t b = b0; b < b1; b++) {
      const int64_t beg = b * c->batch, end = beg + c->batch > n ? n : beg + c->batch;
      orc_fe_warp_batch(c, x, y, t_ns, beg, end, t_ref_ns, omega, iwe, deriv);
    }
  }
#pragma omp parallel for num_threads(T) schedule(static)
  for (size_t i = 0; i < per; i++) {
    float s = ws[i];
    for (int t = 1; t < T; t++) s += ws[per * (size_t)t + i];
    ws[i] = s;
  }
  float *iwe = ws, *deriv = grad ? ws + np : NULL, *tmp = ws + per * (size_t)T;
  if (c->sigma > 0) {
    blur_rows_mt(iwe, tmp, c->W, c->H, 1, c->sigma, T);
    if (deriv) blur_rows_mt(deriv, tmp, c->W, c->H, 3, c->sigma, T);
  }
  const float *ch[3] = {deriv, deriv ? deriv + 1 : NULL, deriv ? deriv + 2 : NULL};
  *contrast = contrast_mt(iwe, np, ch, 3, 3, c->measure, grad, T);
  return 0;
}

/* back end: global_contrast_fdf on T threads (thread-private IL_old / IL_new / derivative planes) */
int orc_be_eval_mt(const orc_be_cfg *c, orc_be_state *st, int64_t n, const uint16_t *x, const uint16_t *y,
                   const int64_t *t_ns, const double *knots0, const double *drotv, int nthreads, double *contrast,
                   double *grad) {
  const size_t np = (size_t)c->Wp * c->Hp;
  const int P = 3 * (c->K - c->num_fixed);
  const size_t per = np * (size_t)(2 + (grad ? P : 0)); /* [IL_old | IL_new | planes] per thread */
  for (int64_t i = 0; i < n; i++)
    if (x[i] >= c->W || y[i] >= c->H) return -1;
  double knots[4 * 64];
  if (c->K > 64) return -3;
  memcpy(knots, knots0, sizeof(double) * 4 * (size_t)c->K);
  for (int i = c->num_fixed; i < c->K; i++) orc_so3_left_update(knots + 4 * i, drotv + 3 * (i - c->num_fixed));
  /* batches as event_pano_warper.cpp:188-196: a trailing single-event batch is skipped */
  int64_t nb = 0;
  for (int64_t beg = 0; beg < n - 1; beg += c->batch) nb++;
  int T = nthreads > 0 ? nthreads : omp_get_max_threads();
  if (T > nb) T = nb > 0 ? (int)nb : 1;
  float *ws = workspace(per * (size_t)T + np);
  if (!ws) return -4;
  int bad = 0;
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    float *mine = ws + per * (size_t)t;
    memset(mine, 0, per * sizeof(float));
    orc_be_state priv = *st;
    priv.IL_old = mine;
    priv.IL_new = mine + np;
    float *planes = grad ? mine + 2 * np : NULL;
    const int64_t b0 = nb * t / T, b1 = nb * (t + 1) / T;
    for (int64_t b = b0; b < b1; b++) {
      const int64_t beg = b * c->batch, left = n - beg;
      const int64_t end = (left > c->batch) ? beg + c->batch : n;
      if (orc_be_warp_batch(c, &priv, x, y, t_ns, beg, end, knots, planes)) {
#pragma omp atomic write
        bad = 1;
      }
    }
  }
  if (bad) return -2;
#pragma omp parallel for num_threads(T) schedule(static)
  for (size_t i = 0; i < per; i++) {
    float s = ws[i];
    for (int t = 1; t < T; t++) s += ws[per * (size_t)t + i];
    ws[i] = s;
  }
  memcpy(st->IL_old, ws, np * sizeof(float));
  memcpy(st->IL_new, ws + np, np * sizeof(float));
  float *planes = grad ? ws + 2 * np : NULL, *iwe = ws + per * (size_t)T;
#pragma omp parallel for num_threads(T) schedule(static)
  for (size_t i = 0; i < np; i++) st->IL[i] = st->IL_old[i] + st->IL_new[i];
  if (st->first_iter) {
    memcpy(st->IGp, st->IG, np * sizeof(float));
    st->alpha = orc_be_alpha(st->IGp, st->IL, (int)np);
    st->first_iter = 0;
  }
  const float a = (float)st->alpha;
#pragma omp parallel for num_threads(T) schedule(static)
  for (size_t i = 0; i < np; i++) iwe[i] = st->IGp[i] * a + st->IL[i];
  if (c->sigma > 0) {
    /* the iwe and every plane are blurred independently: one image per thread, unchanged arithmetic */
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
    for (int k = -1; k < (grad ? P : 0); k++)
      orc_gaussian_blur(k < 0 ? iwe : planes + (size_t)k * np, c->Wp, c->Hp, 1, c->sigma);
  }
  const float *ch[64 * 3];
  for (int k = 0; k < P; k++) ch[k] = planes ? planes + (size_t)k * np : NULL;
  *contrast = contrast_mt(iwe, np, ch, 1, P, c->measure == ORC_MEAN_SQUARE ? ORC_MEAN_SQUARE : ORC_VARIANCE, grad, T);
  return 0;
}
