/*
 * cv_ops.c -- OpenCV / ros::Time semantics the reference's hot path depends on, restated.
 * TEST INFRASTRUCTURE ONLY (see cmax_oracle.h).
 *
 * OpenCV is an un-vendored dependency of the reference (CMakeLists.txt:15, version unpinned; the
 * install doc targets Ubuntu 20.04 / ROS noetic => OpenCV 4.2).  PARITY UNPINNED: nothing in the
 * reference tests any of this.  Restated from OpenCV 4.2's documented behaviour:
 *   cv::GaussianBlur(src,dst,Size(0,0),sigma)  local_image_warped_events.cpp:34-37, event_pano_warper.cpp:219-227
 *   cv::meanStdDev / cv::mean / cv::norm(NORM_L2SQR) / Mat::mul / MatExpr  local_focus_funcs.cpp:9-44,
 *                                                                        global_focus_funcs.cpp:11-47
 *   cv::Sobel  local_focus_funcs.cpp:47-73
 * Compile with -ffp-contract=off: OpenCV's generic (non-FMA) build does separate mul/add.
 */
#include "cmax_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cvRound: round-half-to-even */
static int cv_round(double v) { return (int)lrint(v); }

/* GaussianBlur with ksize=Size(0,0): ksize = cvRound(sigma*(depth==CV_8U?3:4)*2+1)|1 ; CV_32F -> 4 */
int orc_gauss_ksize(double sigma) { return cv_round(sigma * 4 * 2 + 1) | 1; }

/* getGaussianKernel(n, sigma, CV_32F): OpenCV 4.2 computes the kernel in (soft)double, normalises in
 * double, then converts to float. */
void orc_gauss_kernel(int n, double sigma, float *k) {
  double *t = (double *)malloc(sizeof(double) * (size_t)n);
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    t[i] = exp(scale2X * x * x);
    sum += t[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
  free(t);
}

/* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

/* sepFilter2D for CV_32F src/dst/kernels: RowFilter<float,float> (ksize 9 > 5 => generic row filter,
 * s0 = kx[0]*S[0]; s0 += kx[k]*S[k]) then SymmColumnFilter (s0 = ky[c]*S_c; s0 += ky[c+k]*(S_{c+k}+S_{c-k})).
 * In-place use is semantically out-of-place (FilterEngine buffers rows). */
void orc_gaussian_blur(float *img, int W, int H, int cn, double sigma) {
  const int n = orc_gauss_ksize(sigma);
  const int r = n / 2;
  float *kx = (float *)malloc(sizeof(float) * (size_t)n);
  orc_gauss_kernel(n, sigma, kx);
  const size_t rowlen = (size_t)W * cn;
  float *tmp = (float *)malloc(sizeof(float) * rowlen * (size_t)H);
  int *xi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * r));
  for (int i = 0; i < W + 2 * r; i++) xi[i] = reflect101(i - r, W);
  /* row pass */
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
  /* column pass (symmetric form) */
  const float *ky = kx + r;
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
  free(tmp);
  free(kx);
}

/* cv::Sobel(src, dst, CV_32FC1, dx, dy) with ksize=3, scale=1, BORDER_REFLECT_101 */
static void sobel3(const float *src, float *dst, int W, int H, int dx) {
  float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const float a = src[(size_t)y * W + reflect101(x - 1, W)];
      const float b = src[(size_t)y * W + x];
      const float c = src[(size_t)y * W + reflect101(x + 1, W)];
      tmp[(size_t)y * W + x] = dx ? (c - a) : (a + b * 2.f + c);
    }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const float a = tmp[(size_t)reflect101(y - 1, H) * W + x];
      const float b = tmp[(size_t)y * W + x];
      const float c = tmp[(size_t)reflect101(y + 1, H) * W + x];
      dst[(size_t)y * W + x] = dx ? (a + b * 2.f + c) : (c - a);
    }
  free(tmp);
}

/* cv::mean on CV_32F: fp64 accumulation / N */
static double mean_f32(const float *p, size_t n, size_t stride) {
  double s = 0;
  for (size_t i = 0; i < n; i++) s += p[i * stride];
  return s / (double)n;
}

/* contrast_MeanSquare  local_focus_funcs.cpp:9-24, global_focus_funcs.cpp:11-26 */
static double contrast_mean_square(const float *img, int npix, const float *const *ch, int stride, int P,
                                   double *grad) {
  double sq = 0; /* cv::norm(NORM_L2SQR): fp64 accumulation of v*v */
  for (int i = 0; i < npix; i++) sq += (double)img[i] * (double)img[i];
  const double contrast = sq / (double)npix;
  if (grad)
    for (int k = 0; k < P; k++) {
      double s = 0; /* cv::mean(img.mul(ch)) : fp32 product image, fp64 mean */
      for (int i = 0; i < npix; i++) {
        const float m = img[i] * ch[k][(size_t)i * stride];
        s += m;
      }
      grad[k] = 2. * (s / (double)npix);
    }
  return contrast;
}

/* contrast_Variance  local_focus_funcs.cpp:26-44, global_focus_funcs.cpp:29-47 */
static double contrast_variance(const float *img, int npix, const float *const *ch, int stride, int P,
                                double *grad) {
  /* cv::meanStdDev, CV_32F: fp64 sum and sum of squares; var = max(E[x^2]-E[x]^2, 0) (population) */
  double s = 0, sq = 0;
  for (int i = 0; i < npix; i++) {
    const double v = img[i];
    s += v;
    sq += v * v;
  }
  const double mean = s / (double)npix;
  double var = sq / (double)npix - mean * mean;
  if (var < 0) var = 0;
  const double stddev = sqrt(var);
  const double contrast = stddev * stddev; /* :32 */
  if (grad) {
    /* cv::Mat img_zeromean = 2.*(img - mean) -> MatExpr folds to convertTo(CV_32F, alpha=2, beta=-2*mean):
     * fp32 image  z = img*2.f + (float)(-2*mean)  */
    const float beta = (float)(-2. * mean);
    float *z = (float *)malloc(sizeof(float) * (size_t)npix);
    for (int i = 0; i < npix; i++) z[i] = img[i] * 2.f + beta;
    for (int k = 0; k < P; k++) {
      const double mk = mean_f32(ch[k], (size_t)npix, (size_t)stride);
      const float mkf = (float)mk; /* ch - mean(ch): fp32 image, scalar converted to the work type */
      double acc = 0;
      for (int i = 0; i < npix; i++) {
        const float d = ch[k][(size_t)i * stride] - mkf;
        const float m = z[i] * d; /* Mat::mul -> fp32 */
        acc += m;                 /* cv::mean -> fp64 accumulation */
      }
      grad[k] = acc / (double)npix;
    }
    free(z);
  }
  return contrast;
}

/* contrast_ImageGradientMagnitude  local_focus_funcs.cpp:47-73 (front end only; contrast_measure=2 is
 * never set by any launch file) */
static double contrast_gradmag(const float *img, int W, int H, const float *const *ch, int stride, int P,
                               double *grad) {
  const size_t np = (size_t)W * H;
  float *gx = (float *)malloc(sizeof(float) * np), *gy = (float *)malloc(sizeof(float) * np);
  sobel3(img, gx, W, H, 1);
  sobel3(img, gy, W, H, 0);
  double s = 0;
  for (size_t i = 0; i < np; i++) {
    const float hf = gx[i] * gx[i] + gy[i] * gy[i];
    s += hf;
  }
  const double contrast = s / (double)np;
  if (grad) {
    float *c = (float *)malloc(sizeof(float) * np), *dgx = (float *)malloc(sizeof(float) * np),
          *dgy = (float *)malloc(sizeof(float) * np);
    for (int k = 0; k < P; k++) {
      for (size_t i = 0; i < np; i++) c[i] = ch[k][i * (size_t)stride];
      sobel3(c, dgx, W, H, 1);
      sobel3(c, dgy, W, H, 0);
      double a = 0;
      for (size_t i = 0; i < np; i++) {
        const float m = gx[i] * dgx[i] + gy[i] * dgy[i];
        a += m;
      }
      grad[k] = 2. * (a / (double)np);
    }
    free(c); free(dgx); free(dgy);
  }
  free(gx); free(gy);
  return contrast;
}

/* computeContrast  local_focus_funcs.cpp:82-120 / global_focus_funcs.cpp:52-80 */
double orc_contrast(const float *img, int npix, const float *const *ch, int stride, int P, int measure,
                    int W, int H, double *grad) {
  switch (measure) {
    case ORC_MEAN_SQUARE: return contrast_mean_square(img, npix, ch, stride, P, grad);
    case ORC_GRADIENT_MAGNITUDE: return contrast_gradmag(img, W, H, ch, stride, P, grad);
    default: return contrast_variance(img, npix, ch, stride, P, grad);
  }
}

/* ---- ros::Time / ros::Duration (roscpp_core, un-vendored; noetic semantics) ----
 * Duration d = t_last - t_first (exact in ns);  d * 0.5 -> Duration(d.toSec()*0.5) -> fromSec():
 *   sec = floor(x); nsec = round((x - sec)*1e9)      [local_image_warped_events.cpp:68-73,
 *                                                    event_pano_warper.cpp:239-242] */
int64_t orc_time_batch_ns(int64_t t_first_ns, int64_t t_last_ns) {
  const int64_t d = t_last_ns - t_first_ns;
  int64_t dsec = d / 1000000000LL, dnsec = d % 1000000000LL;
  if (dnsec < 0) { dnsec += 1000000000LL; dsec -= 1; } /* normalizeSecNSecSigned */
  const double dsecf = (double)dsec + 1e-9 * (double)dnsec; /* Duration::toSec */
  const double h = dsecf * 0.5;
  const int64_t hs = (int64_t)floor(h);
  int64_t hn = (int64_t)round((h - (double)hs) * 1e9); /* boost::math::round: half away from zero */
  return t_first_ns + hs * 1000000000LL + hn;
}

/* ros::Time::toSec(): (double)sec + 1e-9*(double)nsec */
double orc_time_to_sec(int64_t t_ns) {
  const int64_t sec = t_ns / 1000000000LL, nsec = t_ns % 1000000000LL;
  return (double)sec + 1e-9 * (double)nsec;
}
