// cmx_splat_body.hpp -- the per-workgroup body of the front end's LDS-privatised splat and the vote helpers it shares with the
// back end's (device inline).  Lives in a header because two launches run it: fe_splat_lds_kernel (cmx_binning.hip) and the
// gather + splat launch of the device-driven solve (cmx_kernels.hip).
#pragma once
#include "cmx_internal.hpp"
#include "cmx_warp.hpp"

namespace cmx {

// LDS accumulators are 64-bit fixed point (2^-30 units), not fp32: on gfx950 ds_add_f32 retires ONE lane at a time
// (193 G lane-atomics/s for any address pattern) while ds_add_u64 runs at 1.7 T/s (tools/microbench/lds_atomics.hip).
// Integer adds also commute, so a window's sum does not depend on the order the votes arrive in; the quantisation
// (<= 2^-31 per vote) is far below fp32's own rounding of the reference's accumulators.
typedef unsigned long long fix_t;
constexpr float kFixScale = 1073741824.0f;        // 2^30
constexpr double kFixInv = 1.0 / 1073741824.0;
__device__ __forceinline__ fix_t to_fix(float w) { return (fix_t)(unsigned)(w * kFixScale + 0.5f); }
__device__ __forceinline__ void lds_add_fix(fix_t *p, fix_t v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void vote4_lds(fix_t *win, int lx, int ly, float dx, float dy) {
  fix_t *q = win + ly * kBinStride + lx;
  lds_add_fix(q, to_fix((1.f - dx) * (1.f - dy)));
  lds_add_fix(q + 1, to_fix(dx * (1.f - dy)));
  lds_add_fix(q + kBinStride, to_fix((1.f - dx) * dy));
  lds_add_fix(q + kBinStride + 1, to_fix(dx * dy));
}
__device__ __forceinline__ void vote4_global(float *img, int W, int xx, int yy, float dx, float dy) {
  float *q = img + (size_t)yy * W + xx;
  atomic_add_f32(q, (1.f - dx) * (1.f - dy));
  atomic_add_f32(q + 1, dx * (1.f - dy));
  atomic_add_f32(q + W, (1.f - dx) * dy);
  atomic_add_f32(q + W + 1, dx * dy);
}

// deterministic mode (CMX_OPT_DETERMINISTIC): everything that reaches global memory is a 64-bit INTEGER add into a
// fixed-point plane -- integer adds commute, so the planes (and everything computed from them) are the same bits on every
// run, whatever order the workgroups, the tile sort or the atomics happened in.  fixed_to_float then hands the usual
// fp32 planes to the image kernels and leaves the fixed-point plane all-zero for the next evaluation.
__device__ __forceinline__ void vote4_global_fix(fix_t *img, int W, int xx, int yy, float dx, float dy) {
  fix_t *q = img + (size_t)yy * W + xx;
  atomicAdd(q, to_fix((1.f - dx) * (1.f - dy)));
  atomicAdd(q + 1, to_fix(dx * (1.f - dy)));
  atomicAdd(q + W, to_fix((1.f - dx) * dy));
  atomicAdd(q + W + 1, to_fix(dx * dy));
}

static_assert(kBinWindow * kBinWindow % 256 == 0, "window cells per thread");
constexpr int kUnroll = 2;  // events in flight per thread (swept on MI355X: 2 -> 12.6 us, 1 -> 13.3, 4 -> 14.1, 8 -> 15.2 per 1M events)

// the body of the front-end LDS splat for chunk `chunk` (one workgroup); win: kBinWindow x kBinStride fix_t of LDS, sfall: one
// LDS word.  Shared by fe_splat_lds_kernel (cmx_binning.hip) and the device-driven solve's gather + splat launch (cmx_kernels.hip)
template <bool FIXED, bool STREAM>
__device__ __forceinline__ void fe_splat_lds_body(const FeSplatArgs &a, const BinnedEvents &b, int chunk, fix_t *win, unsigned &sfall) {
  // the launch is sized by an upper bound of the table's length, and so is the table's allocation: the entry is read
  // BEFORE the length is checked, so that the two loads share one memory round trip instead of taking two (~1 us each)
  const Chunk c = b.chunks[chunk];
  if (chunk >= *b.nchunks_dev) return;
  const bool has_win = c.wx0 > -100000000;
  const int tid = threadIdx.x;
  // votes on the global path are counted per workgroup (LDS) and reported with ONE device atomic: a counter every thread
  // adds to is a single memory-side address -- ~1.3 ns per add, 13 us per percent of a million events' votes
  if (tid == 0) sfall = 0;
  if (has_win)
    for (int p = tid; p < kBinWindow * kBinStride; p += 256) win[p] = 0ull;
  __syncthreads();
  unsigned nfall = 0;
  for (int j0 = c.beg + tid; j0 < c.end; j0 += 256 * kUnroll) {
    bool act[kUnroll];
    double px[kUnroll], py[kUnroll], pz[kUnroll], dt[kUnroll];
    if (STREAM) {  // bearing and dt of every sorted event stream in (coalesced): no table gathers in this kernel
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const int j = j0 + u * 256;
        act[u] = j < c.end;
        const int jj = act[u] ? j : c.beg;
        const double2 v = *reinterpret_cast<const double2 *>(b.sb + 2 * (size_t)jj);
        px[u] = v.x; py[u] = v.y; pz[u] = 1.0;
        dt[u] = b.sdt[jj];
      }
    } else {
      uint32_t e[kUnroll], bi[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const int j = j0 + u * 256;
        act[u] = j < c.end;
        e[u] = act[u] ? b.sxy[j] : 0u;
        bi[u] = act[u] ? b.sbatch[j] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        load_bearing(a, (int)(e[u] & 0xffff), (int)((e[u] >> 16) & 0x7fff), px[u], py[u], pz[u]);
        dt[u] = a.batch_dt[bi[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      const FeWarp w = fe_warp_math<false>(a, px[u], py[u], pz[u], dt[u]);
      if (act[u] && w.ok) {
        const int lx = w.xx - c.wx0, ly = w.yy - c.wy0;
        if (has_win && lx >= 0 && lx < kBinWindow - 1 && ly >= 0 && ly < kBinWindow - 1) {
          vote4_lds(win, lx, ly, w.dx, w.dy);
        } else {
          if (FIXED) vote4_global_fix(b.fixed, a.W, w.xx, w.yy, w.dx, w.dy);
          else vote4_global(a.planes, a.W, w.xx, w.yy, w.dx, w.dy);
          nfall++;
        }
      }
    }
  }
  if (nfall) atomicAdd(&sfall, nfall);
  __syncthreads();
  if (tid == 0 && sfall) atomicAdd(b.fallback, sfall);
  if (has_win) {
    // all of a thread's window cells are read before the first is flushed: one LDS round trip instead of sixteen
    // (the rolled loop waited for every read in turn: ~0.9 of the kernel's ~9 us, profiles/r02_splat_timeline.txt)
    constexpr int kCells = kBinWindow * kBinWindow / 256;
    fix_t cell[kCells];
#pragma unroll
    for (int k = 0; k < kCells; k++) {
      const int p = tid + 256 * k, ly = p / kBinWindow, lx = p - ly * kBinWindow;
      cell[k] = win[ly * kBinStride + lx];
    }
#pragma unroll
    for (int k = 0; k < kCells; k++) {
      const fix_t v = cell[k];
      if (v != 0ull) {
        const int p = tid + 256 * k, ly = p / kBinWindow, lx = p - ly * kBinWindow;
        const size_t at = (size_t)(c.wy0 + ly) * a.W + (c.wx0 + lx);
        if (FIXED) atomicAdd(b.fixed + at, v);
        else atomic_add_f32(a.planes + at, (float)((double)v * kFixInv));
      }
    }
  }
}

}  // namespace cmx
