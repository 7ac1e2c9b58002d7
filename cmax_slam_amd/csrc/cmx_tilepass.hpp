// cmx_tilepass.hpp -- the adjoint image pass of ONE strip (32 x 16 pixels) of a 32 x 32 sort tile, run by a workgroup of the front-end
// LDS splat's launch the moment the strip's inputs are complete (tile-dataflow fusion, FusedArgs in cmx_internal.hpp; round 6).
//
// Same arithmetic as image_adjoint2_kernel<4, ..> (cmx_kernels.hip), pixel for pixel:
//   raw (tile + 2r)  ->  [row pass: G_x raw on tile rows +-r, M_x raw on tile rows +-2r]  ->  [column pass: B and Jt on the tile]
// B = G I in the operation order of image_moments_kernel (plain fp32 multiply / add, no contraction), Jt = My (Mx I) with the
// 17-term sums of the banded operator M = G^T G accumulated in fp64 (reference: cv::GaussianBlur of the IWE,
// local_image_warped_events.cpp:32-38, and the variance / gradient of local_focus_funcs.cpp:26-44 in its adjoint form,
// DESIGN.md section 4.2).  Only the grouping of the two image moments differs (32 x 16 strips instead of 64 x 16 tiles).
//
// Memory model.  The raw pixels were written by OTHER workgroups of the SAME launch, on any XCD, with agent-scope atomic adds
// (performed at the coherent level), and every one of those workgroups drained its atomics (s_waitcnt vmcnt(0)) before its
// arrival on the tile's counter -- so the completing arrival may read them, provided it does not hit a stale line of its own
// XCD's L2 / its CU's L1: the loads are agent-scope (sc1) loads, the same rule as the tail finalize (cmx_kernels.hip,
// tail_arrive: "sc1 loads may replace the acquire only when the producer stored sc1").  Everything the pass writes (Jt, the
// partner's cleared tile, moment rows) is read by LATER launches.
#pragma once
#include "cmx_internal.hpp"

namespace cmx {

__device__ __forceinline__ int tp_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
__device__ __forceinline__ float tp_ld_sc1(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

constexpr int kTpR = 4, kTpT = kBinTile, kTpH = kBinTile / kFuseStrips;  // a strip: kTpT columns x kTpH lines
constexpr int kTpAW = kTpT + 4 * kTpR, kTpAH = kTpH + 4 * kTpR, kTpGH = kTpH + 2 * kTpR, kTpTap = 4 * kTpR + 1;
// LDS of one pass: reduction scratch | raw (float) | G_x raw (float) | M_x raw (double) | the strip's rows of M_y (double)
constexpr size_t kTpLdsBytes = sizeof(double) * 32 + sizeof(float) * ((size_t)kTpAW * kTpAH + (size_t)kTpT * kTpGH) +
                               sizeof(double) * ((size_t)kTpT * kTpAH + (size_t)kTpH * kTpTap);
constexpr int kTpRawPerThread = (kTpAW * kTpAH + 511) / 512;  // raw pixels per thread of a 512-thread workgroup

// N agent-scope (sc1) dword loads in flight at once, then ONE wait.  As relaxed atomic loads the compiler waits for each of them
// in turn (dependent ~1-2 us trips to memory per tile); the loads and the wait therefore live in one asm statement.
__device__ __forceinline__ void tp_ld_sc1_n(const float *const (&p)[3], float (&v)[3]) {
  asm volatile(
      "global_load_dword %0, %3, off sc1\n\t"
      "global_load_dword %1, %4, off sc1\n\t"
      "global_load_dword %2, %5, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
      : "v"(p[0]), "v"(p[1]), "v"(p[2])
      : "memory");
}
__device__ __forceinline__ void tp_ld_sc1_n(const float *const (&p)[5], float (&v)[5]) {
  asm volatile(
      "global_load_dword %0, %5, off sc1\n\t"
      "global_load_dword %1, %6, off sc1\n\t"
      "global_load_dword %2, %7, off sc1\n\t"
      "global_load_dword %3, %8, off sc1\n\t"
      "global_load_dword %4, %9, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4])
      : "memory");
}

// NT threads (a multiple of kTpT, at least 64), all of them call; `lds` = kTpLdsBytes bytes, 16-byte aligned, free for this call.
// wait_inputs(): called by every thread once everything that does NOT depend on the tile's votes has been requested (operator
// rows, taps) -- it returns (in every thread, behind a workgroup barrier) whether the votes are complete; false: nothing is computed.
// The caller issues a barrier before it reuses `lds`.
// PUB: the pass's outputs (Jt, the moment row) are read by OTHER workgroups of the same launch (the one-launch evaluation's gather
// role and finalize): stored write-through (agent scope), the caller publishes the strip's done flag behind them.
template <int NT, bool PUB, typename WaitFn>
__device__ __forceinline__ void fused_tile_pass(const FusedArgs &f, const float *plane, int W, int H, int strip, unsigned char *lds,
                                                WaitFn wait_inputs) {
  constexpr int R = kTpR, T = kTpT, TH = kTpH, AW = kTpAW, AH = kTpAH, GH = kTpGH, NTAP = kTpTap, NRAW = kTpRawPerThread;
  static_assert(NT % T == 0 && NT >= 64 && NT % 64 == 0, "tile pass geometry");
  static_assert(AW * AH <= NRAW * NT, "raw pixels per thread");
  constexpr int ROWS = NT / T;  // strip lines one sweep of the workgroup covers
  double *red = reinterpret_cast<double *>(lds);
  float *bufA = reinterpret_cast<float *>(lds + 32 * sizeof(double));  // raw, AW x AH
  float *bufG = bufA + AW * AH;                                        // G_x raw, T x GH (strip lines -r .. TH+r)
  double *bufM = reinterpret_cast<double *>(bufG + T * GH);            // M_x raw (rounded to fp32, kept as fp64), T x AH (lines -2r .. TH+2r)
  double *bufY = bufM + T * AH;                                        // rows of M_y for the strip's TH lines, TH x NTAP, as fp64
  static_assert((sizeof(double) * 32 + sizeof(float) * (AW * AH + T * GH)) % 8 == 0, "fp64 buffers 8-byte aligned");
  const int tid = threadIdx.x, tx = tid % T, ty0 = tid / T;
  const int tile = strip / kFuseStrips;
  const int x0 = (tile % f.tiles_x) * T, y0 = (tile / f.tiles_x) * T + (strip % kFuseStrips) * TH;
  if (y0 >= H) {  // (the lower strips of a partial bottom tile row lie outside the image: nothing to compute, but the counter is consumed)
    (void)wait_inputs();
    return;
  }
  // ---- what does not depend on the votes: taps, this column's row of M_x (registers), the strip's rows of M_y (LDS) -- the
  // operator rows as fp64: v_cvt_f64_f32 is a quarter-rate instruction, and with both factors of every one of the pass's 2 x 17
  // products per pixel converted where they are used the conversions were most of the pass (2.7 us of LDS / ALU phases per tile)
  float taps[2 * R + 1];
#pragma unroll
  for (int j = 0; j < 2 * R + 1; j++) taps[j] = f.taps[j];
  double mx[NTAP];
  {
    const bool interior = x0 >= 2 * R && x0 + T - 1 <= W - 1 - 2 * R;
    const float *row = f.Mx + (size_t)(interior ? 2 * R : min(x0 + tx, W - 1)) * NTAP;
#pragma unroll
    for (int i = 0; i < NTAP; i++) mx[i] = (double)row[i];
  }
  {
    const bool interior = y0 >= 2 * R && y0 + TH - 1 <= H - 1 - 2 * R;
    for (int idx = tid; idx < TH * NTAP; idx += NT) {
      const int ty = idx / NTAP, i = idx - ty * NTAP;
      bufY[idx] = (double)f.My[(size_t)(interior ? 2 * R : min(y0 + ty, H - 1)) * NTAP + i];
    }
  }
  // raw strip + 2r halo (REFLECT_101 beyond the image for G_x; M's rows carry zeros there): addresses first
  const float *rp[NRAW];
#pragma unroll
  for (int k = 0; k < NRAW; k++) {
    const int idx = min(tid + k * NT, AW * AH - 1);
    const int ly = idx / AW, lx = idx - ly * AW;
    const int gx = tp_reflect101(x0 + lx - 2 * R, W), gy = tp_reflect101(y0 + ly - 2 * R, H);
    rp[k] = plane + (size_t)gy * W + gx;
  }
  if (!wait_inputs()) return;
  {
    float v[NRAW];
    tp_ld_sc1_n(rp, v);
#pragma unroll
    for (int k = 0; k < NRAW; k++)
      if (tid + k * NT < AW * AH) bufA[tid + k * NT] = v[k];
  }
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 4] = wall_clock64();
  if (f.zero_ptr) {  // ping-pong: the previous evaluation's votes on this strip (nobody reads that buffer in this launch)
    for (int idx = tid; idx < T * TH; idx += NT) {
      const int gx = x0 + (idx % T), gy = y0 + (idx / T);
      if (gx < W && gy < H) f.zero_ptr[(size_t)gy * W + gx] = 0.f;
    }
  }
  __syncthreads();
  for (int ly = ty0; ly < AH; ly += ROWS) {  // row pass: raw row ly, output column tx
    const float *S = bufA + ly * AW + tx;
    float in[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; i++) in[i] = S[i];
    // four independent fp64 chains (i mod 4) instead of one of 17 dependent FMAs: with two waves per SIMD nothing hides a dependent
    // v_fma_f64's latency.  (The sum's grouping differs from image_adjoint2's: ~1e-16 relative, before the rounding to fp32.)
    double mc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) mc[q] = mx[q] * (double)in[q];
#pragma unroll
    for (int i = 4; i < NTAP; i++) mc[i & 3] = __builtin_fma(mx[i], (double)in[i], mc[i & 3]);
    const double m = (mc[0] + mc[1]) + (mc[2] + mc[3]);
    bufM[ly * T + tx] = (double)(float)m;  // (rounded to fp32 as image_adjoint2 stores it)
    if (ly >= R && ly < R + GH) {  // forward row pass, same op order as image_moments
      float s = taps[0] * in[R];
#pragma unroll
      for (int j = 1; j <= 2 * R; j++) s += taps[j] * in[R + j];
      bufG[(ly - R) * T + tx] = s;
    }
  }
  __syncthreads();
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 5] = wall_clock64();
  double sI = 0, sII = 0;
  for (int ty = ty0; ty < TH; ty += ROWS) {  // column pass: B (moments) and Jt at (x0 + tx, y0 + ty)
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
      const float *Tg = bufG + (ty + R) * T + tx;
      float s = taps[R] * Tg[0];
#pragma unroll
      for (int t = 1; t <= R; t++) s += taps[R + t] * (Tg[t * T] + Tg[-t * T]);
      sI += (double)s;
      sII += (double)s * (double)s;
      const double *my = bufY + ty * NTAP;  // (wave-uniform per half wave: LDS broadcast)
      const double *Q = bufM + ty * T + tx;
      double jc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) jc[q] = my[q] * Q[q * T];
#pragma unroll
      for (int i = 4; i < NTAP; i++) jc[i & 3] = __builtin_fma(my[i], Q[i * T], jc[i & 3]);
      const double j = (jc[0] + jc[1]) + (jc[2] + jc[3]);
      if (PUB) __hip_atomic_store(reinterpret_cast<unsigned *>(f.jt + (size_t)gy * W + gx), __float_as_uint((float)j), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
      else f.jt[(size_t)gy * W + gx] = (float)j;
    }
  }
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 6] = wall_clock64();
  // the strip's two moments: wave sums, one LDS slot per wave, thread 0 adds them in wave order
  {
    double v0 = sI, v1 = sII;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      v0 += __shfl_xor(v0, o, 64);
      v1 += __shfl_xor(v1, o, 64);
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { red[2 * wave] = v0; red[2 * wave + 1] = v1; }
    __syncthreads();
    if (tid == 0) {
      double t0 = 0, t1 = 0;
      for (int w = 0; w < NT / 64; w++) { t0 += red[2 * w]; t1 += red[2 * w + 1]; }
      if (f.macc) {  // device-driven solve: accumulator rows read by every workgroup of the gradient pass queued behind this launch
        double *row = f.macc + (size_t)(strip % kTailShards) * 16;
        if (t0 != 0.0) __hip_atomic_fetch_add(row, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t1 != 0.0) __hip_atomic_fetch_add(row + 1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int nstrips = f.tiles_x * f.tiles_y * kFuseStrips;
        if (PUB) {
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.partials + strip), (unsigned long long)__double_as_longlong(t0),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.partials + nstrips + strip), (unsigned long long)__double_as_longlong(t1),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          f.partials[strip] = t0;
          f.partials[nstrips + strip] = t1;
        }
      }
    }
  }
}

}  // namespace cmx
