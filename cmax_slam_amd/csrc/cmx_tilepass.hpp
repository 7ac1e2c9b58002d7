// cmx_tilepass.hpp -- the adjoint image pass of ONE 32 x 32 tile, run by a chunk workgroup of the front-end LDS splat the moment
// the tile's inputs are complete (tile-dataflow fusion, FusedArgs in cmx_internal.hpp; round 6).
//
// Same arithmetic as image_adjoint2_kernel<4, ..> (cmx_kernels.hip), pixel for pixel:
//   raw (tile + 2r)  ->  [row pass: G_x raw on tile rows +-r, M_x raw on tile rows +-2r]  ->  [column pass: B and Jt on the tile]
// B = G I in the operation order of image_moments_kernel (plain fp32 multiply / add, no contraction), Jt = My (Mx I) with the
// 17-term sums of the banded operator M = G^T G accumulated in fp64 (reference: cv::GaussianBlur of the IWE,
// local_image_warped_events.cpp:32-38, and the variance / gradient of local_focus_funcs.cpp:26-44 in its adjoint form,
// DESIGN.md section 4.2).  Only the grouping of the two image moments differs (32 x 32 tiles instead of 64 x 16).
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

constexpr int kTpR = 4, kTpT = kBinTile, kTpA = kTpT + 4 * kTpR, kTpG = kTpT + 2 * kTpR, kTpTap = 4 * kTpR + 1;
constexpr size_t kTpLdsBytes =
    sizeof(float) * ((size_t)kTpA * kTpA + (size_t)kTpT * kTpG + (size_t)kTpT * kTpA + (size_t)kTpT * kTpTap) + sizeof(double) * 32;

// five agent-scope (sc1) dword loads in flight at once, then ONE wait.  As relaxed atomic loads the compiler waits for each of
// them in turn (five dependent ~2 us trips to memory per tile); the loads and the wait therefore live in one asm statement.
__device__ __forceinline__ void tp_ld5_sc1(const float *p0, const float *p1, const float *p2, const float *p3, const float *p4,
                                           float &v0, float &v1, float &v2, float &v3, float &v4) {
  asm volatile(
      "global_load_dword %0, %5, off sc1\n\t"
      "global_load_dword %1, %6, off sc1\n\t"
      "global_load_dword %2, %7, off sc1\n\t"
      "global_load_dword %3, %8, off sc1\n\t"
      "global_load_dword %4, %9, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4)
      : "memory");
}

// NT threads (a multiple of kTpT, at least 64), all of them call; `lds` = kTpLdsBytes bytes, 16-byte aligned, free for this call.
// wait_inputs(): called by every thread once everything that does NOT depend on the tile's votes has been requested (operator
// rows, taps) -- it returns (in every thread, behind a workgroup barrier) whether the votes are complete; false: nothing is computed.
// The caller issues a barrier before it reuses `lds`.
template <int NT, typename WaitFn>
__device__ __forceinline__ void fused_tile_pass(const FusedArgs &f, const float *plane, int W, int H, int tile, unsigned char *lds,
                                                WaitFn wait_inputs) {
  constexpr int R = kTpR, T = kTpT, AW = kTpA, AH = kTpA, GH = kTpG, NTAP = kTpTap;
  static_assert(NT % T == 0 && NT >= 64 && NT % 64 == 0, "tile pass geometry");
  static_assert(AW * AH <= 5 * NT, "five raw pixels per thread");
  constexpr int ROWS = NT / T;  // tile rows one sweep of the workgroup covers
  double *red = reinterpret_cast<double *>(lds);
  float *bufA = reinterpret_cast<float *>(lds + 32 * sizeof(double));  // raw, AW x AH
  float *bufG = bufA + AW * AH;                                        // G_x raw, T x GH (tile rows -r .. T+r)
  float *bufM = bufG + T * GH;                                         // M_x raw, T x AH (tile rows -2r .. T+2r)
  float *bufY = bufM + T * AH;                                         // rows of M_y for the tile's T lines, T x NTAP
  const int tid = threadIdx.x, tx = tid % T, ty0 = tid / T;
  const int x0 = (tile % f.tiles_x) * T, y0 = (tile / f.tiles_x) * T;
  // ---- what does not depend on the votes: taps, this column's row of M_x (registers), the tile's rows of M_y (LDS)
  float taps[2 * R + 1];
#pragma unroll
  for (int j = 0; j < 2 * R + 1; j++) taps[j] = f.taps[j];
  float mx[NTAP];
  {
    const bool interior = x0 >= 2 * R && x0 + T - 1 <= W - 1 - 2 * R;
    const float *row = f.Mx + (size_t)(interior ? 2 * R : min(x0 + tx, W - 1)) * NTAP;
#pragma unroll
    for (int i = 0; i < NTAP; i++) mx[i] = row[i];
  }
  {
    const bool interior = y0 >= 2 * R && y0 + T - 1 <= H - 1 - 2 * R;
    for (int idx = tid; idx < T * NTAP; idx += NT) {
      const int ty = idx / NTAP, i = idx - ty * NTAP;
      bufY[idx] = f.My[(size_t)(interior ? 2 * R : min(y0 + ty, H - 1)) * NTAP + i];
    }
  }
  // raw tile + 2r halo (REFLECT_101 beyond the image for G_x; M's rows carry zeros there): addresses first
  const float *rp[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int idx = min(tid + k * NT, AW * AH - 1);
    const int ly = idx / AW, lx = idx - ly * AW;
    const int gx = tp_reflect101(x0 + lx - 2 * R, W), gy = tp_reflect101(y0 + ly - 2 * R, H);
    rp[k] = plane + (size_t)gy * W + gx;
  }
  if (!wait_inputs()) return;
  {
    float v[5];
    tp_ld5_sc1(rp[0], rp[1], rp[2], rp[3], rp[4], v[0], v[1], v[2], v[3], v[4]);
#pragma unroll
    for (int k = 0; k < 5; k++)
      if (tid + k * NT < AW * AH) bufA[tid + k * NT] = v[k];
  }
  if (f.zero_ptr) {  // ping-pong: the previous evaluation's votes on this tile (nobody reads that buffer in this launch)
    for (int idx = tid; idx < T * T; idx += NT) {
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
    double m = (double)mx[0] * (double)in[0];
#pragma unroll
    for (int i = 1; i < NTAP; i++) m = __builtin_fma((double)mx[i], (double)in[i], m);
    bufM[ly * T + tx] = (float)m;
    if (ly >= R && ly < R + GH) {  // forward row pass, same op order as image_moments
      float s = taps[0] * in[R];
#pragma unroll
      for (int j = 1; j <= 2 * R; j++) s += taps[j] * in[R + j];
      bufG[(ly - R) * T + tx] = s;
    }
  }
  __syncthreads();
  double sI = 0, sII = 0;
  for (int ty = ty0; ty < T; ty += ROWS) {  // column pass: B (moments) and Jt at (x0 + tx, y0 + ty)
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
      const float *Tg = bufG + (ty + R) * T + tx;
      float s = taps[R] * Tg[0];
#pragma unroll
      for (int t = 1; t <= R; t++) s += taps[R + t] * (Tg[t * T] + Tg[-t * T]);
      sI += (double)s;
      sII += (double)s * (double)s;
      const float *my = bufY + ty * NTAP;  // (wave-uniform per half wave: LDS broadcast)
      const float *Q = bufM + ty * T + tx;
      double j = (double)my[0] * (double)Q[0];
#pragma unroll
      for (int i = 1; i < NTAP; i++) j = __builtin_fma((double)my[i], (double)Q[i * T], j);
      f.jt[(size_t)gy * W + gx] = (float)j;
    }
  }
  // the tile's two moments: wave sums, one LDS slot per wave, thread 0 adds them in wave order
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
        double *row = f.macc + (size_t)(tile % kTailShards) * 16;
        if (t0 != 0.0) __hip_atomic_fetch_add(row, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t1 != 0.0) __hip_atomic_fetch_add(row + 1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int ntiles = f.tiles_x * f.tiles_y;
        f.partials[tile] = t0;
        f.partials[ntiles + tile] = t1;
      }
    }
  }
}

}  // namespace cmx
