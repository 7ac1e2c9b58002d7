// cmx_internal.hpp -- kernel parameter blocks and launcher prototypes shared by cmx_kernels.hip and
// cmx_capi.cpp.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cmx_so3.hpp"

namespace cmx {

constexpr int kMaxRadius = 12;  // Gaussian half-width supported by the fused blur kernel (sigma <= 3)
constexpr int kTileX = 64, kTileY = 16, kImgThreads = 256, kPlaneGroup = 4;

// events are packed  x | y << 16 | old_flag << 31   (old_flag: ev.ts < t_next_win_beg_, back end only)
struct FeSplatArgs {
  double fx, fy, cx, cy;
  double wx, wy, wz;
  int W, H;
  int per_batch;  // events per batch in the packed list
  int n;          // packed events
  const uint32_t *xy;
  const double *batch_dt;  // per batch: time_batch.toSec() - time_ref.toSec()
  const double *lut;       // W*H*3
  float *planes;           // [1 + 3][H][W]: IWE, dI/dwx, dI/dwy, dI/dwz
};

struct PoseEntry {  // per event batch, written by the pose-table kernel
  double R[9];      // so3.matrix(), row-major
  float Jcp[36];    // ddrot_ddrot_cp, 3 x 3n row-major (n<=4)
  int idx_cp_beg;
  int pad;
};

struct BeSplatArgs {
  int W;          // sensor width (LUT index)
  int Wp, Hp;
  double fx, fy, cxp, cyp;  // equirectangular scale / centre
  int per_batch, n;
  int order;      // 2 / 4
  int num_fixed;
  const uint32_t *xy;
  const PoseEntry *poses;
  const double *lut;
  float *planes;  // [2 + P][Hp][Wp]: IL_old, IL_new, derivative planes
};

struct ImgArgs {
  int W, H, r;
  float taps[2 * kMaxRadius + 1];
  // plane 0 = (igp ? igp*alpha : 0) + (src_a + (src_b ? src_b : 0))
  const float *src_a, *src_b, *igp;
  const double *alpha;  // device scalar (back end) or nullptr
  const float *dplanes; // P derivative planes, plane stride = W*H
  int P;
  float *out_blur0;     // optional: blurred plane 0
  float *out_blurd;     // optional: blurred derivative planes
  double *partials;     // [2 + 2P][nblk]
  int nblk;             // tiles in x*y
  int tiles_x;
};

struct FinalizeArgs {
  int P, nblk, measure;
  double npix;
  const double *partials;  // [2+2P][nblk]
  double *sums;            // [2+2P] device scratch
  double *result;          // mapped host: [0]=contrast, [1]=mean, [2..2+P) = gradient
};

struct AlphaArgs {
  const float *igp, *il_old, *il_new;
  int npix;
  double *partials;  // [5][nblk]
  int nblk;
  double *alpha;     // device scalar out
  double *result_alpha;  // mapped host copy
};

void launch_fe_splat(const FeSplatArgs &a, bool deriv, hipStream_t s);
void launch_be_pose_table(const SplineArgs *d_spline, const long long *d_batch_t, int nb, int order, bool want_j,
                          PoseEntry *out, hipStream_t s);
void launch_be_splat(const BeSplatArgs &a, bool deriv, hipStream_t s);
void launch_image_moments(const ImgArgs &a, hipStream_t s);
void launch_finalize(const FinalizeArgs &a, hipStream_t s);
void launch_alpha(const AlphaArgs &a, hipStream_t s);
void launch_interleave3(const float *planes, float *out, int npix, hipStream_t s);
size_t image_lds_bytes(int r);

}  // namespace cmx
