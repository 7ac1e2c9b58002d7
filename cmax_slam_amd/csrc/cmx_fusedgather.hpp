// cmx_fusedgather.hpp -- the GATHER role and the finalize step of the one-launch front-end evaluation
// (fe_splat_lds_kernel<.., FUSE = 2>, FusedArgs in cmx_internal.hpp; round 6).
//
// What it computes is fe_gather_kernel's sum (cmx_kernels.hip; the reference scatters the same signed bilinear weights into its
// derivative images, local_image_warped_events.cpp:157-166, and reduces them in local_focus_funcs.cpp:36-40):
//   S1_k = sum_events  r0_k * dJt/dx + r1_k * dJt/dy   at the event's vote cell,   S2_k = the same against c = G^T 1 (border band),
//   gradient_k = (2/N) (S1_k - mu * S2_k)              (variance; mean square: (2/N) S1_k)
// with the event's warp (fp64, cmx_warp.hpp fe_warp_math -- bit for bit the splat's, so the vote cell is the splat's) done BEFORE
// Jt exists: a gather workgroup is resident while the chunks are still voting, and everything but the four Jt cells per event is
// ready in registers when the strips that own those cells report in.
#pragma once
#include "cmx_internal.hpp"
#include "cmx_warp.hpp"

namespace cmx {

constexpr int kFgU = 4;  // events per thread (kept in registers across the wait: cell offset, dx, dy, 2 x 3 Jacobian rows)

// 16 agent-scope dword loads (the four Jt cells of four events), all in flight, ONE wait.  off[u]: byte offset of cell (yy, xx);
// row0 = Jt, row1 = Jt + one image row.  (saddr form: one 32-bit offset register per event.)
__device__ __forceinline__ void fg_ld16_sc1(const float *row0, const float *row1, const unsigned (&off)[kFgU], float (&c)[kFgU][4]) {
  asm volatile(
      "global_load_dword %0, %16, %20 sc1\n\t"
      "global_load_dword %1, %16, %20 offset:4 sc1\n\t"
      "global_load_dword %2, %16, %21 sc1\n\t"
      "global_load_dword %3, %16, %21 offset:4 sc1\n\t"
      "global_load_dword %4, %17, %20 sc1\n\t"
      "global_load_dword %5, %17, %20 offset:4 sc1\n\t"
      "global_load_dword %6, %17, %21 sc1\n\t"
      "global_load_dword %7, %17, %21 offset:4 sc1\n\t"
      "global_load_dword %8, %18, %20 sc1\n\t"
      "global_load_dword %9, %18, %20 offset:4 sc1\n\t"
      "global_load_dword %10, %18, %21 sc1\n\t"
      "global_load_dword %11, %18, %21 offset:4 sc1\n\t"
      "global_load_dword %12, %19, %20 sc1\n\t"
      "global_load_dword %13, %19, %20 offset:4 sc1\n\t"
      "global_load_dword %14, %19, %21 sc1\n\t"
      "global_load_dword %15, %19, %21 offset:4 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[0][2]), "=&v"(c[0][3]), "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[1][2]), "=&v"(c[1][3]),
        "=&v"(c[2][0]), "=&v"(c[2][1]), "=&v"(c[2][2]), "=&v"(c[2][3]), "=&v"(c[3][0]), "=&v"(c[3][1]), "=&v"(c[3][2]), "=&v"(c[3][3])
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(row0), "s"(row1)
      : "memory");
}

__device__ __forceinline__ double fg_ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void fg_st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fg_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct FgSmem {
  double red[8 * 6];      // per-wave partial sums of the six gradient columns
  double mom[2 * 8];      // finalize: per-wave partial sums of the two image moments
  double cols[8];         // finalize: the six column sums
  double outv[8];
  unsigned long long chk;
  unsigned tile_bits[128];  // tiles this workgroup's vote cells lie in (<= 4096 sort tiles)
  int pending, is_last, ok;
};

// the finalize step of the one-launch evaluation (one workgroup, NT threads): the arithmetic of finalize_body (cmx_kernels.hip)
// for the front end's adjoint gradient with accumulator rows -- contrast from the strips' moment rows, grad = (2/N)(S1 - mu S2)
template <int NT>
__device__ __forceinline__ void fg_finalize(const FusedArgs &f, unsigned *fallback, FgSmem &sm) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int nstrips = f.tiles_x * f.tiles_y * kFuseStrips;
  // every strip of this launch must have stored its moments: usually long true (the gather workgroups waited for the strips their
  // votes touch; strips that only see a neighbour's blur may still be running).  The independent reads -- moment rows (agent-scope
  // loads: stored write-through by the strips), accumulator rows, fallback word -- are requested TOGETHER with the first read of the
  // strip count (one round trip instead of two); only if that count was short are the moment rows read again behind the wait.
  const unsigned want = (unsigned)*f.n_active;
  unsigned done0 = 0u;
  if (t == 0) {
    done0 = __hip_atomic_load(f.tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sm.chk = 0ull;
  }
  double p0 = 0, p1 = 0;
  for (int b = t; b < nstrips; b += NT) {
    p0 += fg_ld_sc1(f.partials + b);
    p1 += fg_ld_sc1(f.partials + nstrips + b);
  }
  double gv[kTailShards];
  if (t < 6) {
#pragma unroll
    for (int q = 0; q < kTailShards; q++) gv[q] = fg_ld_sc1(f.gacc + (size_t)q * f.gacc_stride + t);
  }
  if (t == 0) {
    int ok = 1, late = 0;
    if (done0 < want) {
      late = 1;
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(f.tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000ull) { ok = 0; break; }
      }
    }
    __hip_atomic_store(f.tiles_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ok) atomicOr(fallback, kFuseIncomplete);
    sm.pending = late;
  }
  __syncthreads();
  if (sm.pending) {  // (workgroup-uniform) strips were still running when the rows were read: once more, behind the wait
    p0 = p1 = 0;
    for (int b = t; b < nstrips; b += NT) {
      p0 += fg_ld_sc1(f.partials + b);
      p1 += fg_ld_sc1(f.partials + nstrips + b);
    }
  }
  unsigned fb = 0u;
  if (t == 0) fb = __hip_atomic_load(fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p0 = fg_wave_sum(p0);
  p1 = fg_wave_sum(p1);
  if (lane == 0) { sm.mom[2 * wave] = p0; sm.mom[2 * wave + 1] = p1; }
  if (t < 6) {
    double w = 0;
#pragma unroll
    for (int q = 0; q < kTailShards; q++) w += gv[q];
    sm.cols[t] = w;
#pragma unroll
    for (int q = 0; q < kTailShards; q++) fg_st_sc1(f.gacc + (size_t)q * f.gacc_stride + t, 0.0);  // all-zero again for the next launch
  }
  __syncthreads();
  if (t == 0) {
    double s0 = 0, s1 = 0;
    for (int w = 0; w < NT / 64; w++) { s0 += sm.mom[2 * w]; s1 += sm.mom[2 * w + 1]; }
    double mu;
    const double c = contrast_from_sums(s0, s1, f.npix, f.measure, &mu);
    sm.outv[0] = c;
    sm.outv[1] = mu;
    for (int k = 0; k < 3; k++) sm.outv[2 + k] = 2.0 * (sm.cols[k] - (f.measure != 1 ? mu * sm.cols[3 + k] : 0.0)) / f.npix;
    sm.outv[5] = (double)fb;
    __hip_atomic_store(fallback, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // results, checksum, ticket (the host accepts a snapshot only when ticket AND checksum match: spin_for_ticket, cmx_pipeline.cpp)
  constexpr int nout = 5;
  unsigned long long bits = 0ull;
  if (t < nout) {
    const double v = sm.outv[t];
    f.result[t] = v;
    bits = (unsigned long long)__double_as_longlong(v);
  } else if (t == nout) {
    const double v = sm.outv[5];
    f.result[kFallbackSlot] = v;
    bits = (unsigned long long)__double_as_longlong(v);
  }
  if (t <= nout) atomicXor(&sm.chk, bits);
  __syncthreads();
  if (t == 0) {
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(f.result);
    slots[kChecksumSlot] = sm.chk ^ (f.ticket * kTicketMix);
    slots[kTicketSlot] = f.ticket;
  }
}

// NT threads; g = this gather workgroup's index (0 .. f.gather_blocks - 1)
template <int NT>
__device__ __forceinline__ void fused_gather_role(const FeSplatArgs &a, const BinnedEvents &b, const FusedArgs &f, int g, FgSmem &sm) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int W = a.W, H = a.H, r = kTpR;
  const int blk_beg = g * f.gather_per_block, blk_end = min(a.n, blk_beg + f.gather_per_block);
  for (int k = tid; k < 128; k += NT) sm.tile_bits[k] = 0u;
  if (tid == 0) sm.pending = 0;
  if (f.trace && tid == 0) { f.trace[8 * (size_t)blockIdx.x] = wall_clock64(); f.trace[8 * (size_t)blockIdx.x + 3] = 4; }
  // ---- phase 1 (no dependence on this launch's votes): streams, warp, Jacobian rows, border term
  bool ok[kFgU];
  unsigned off[kFgU];
  float dx[kFgU], dy[kFgU], r0[kFgU][3], r1[kFgU][3];
  double acc2[3] = {0, 0, 0};
  {
    double2 bv[kFgU];
    double dt[kFgU];
#pragma unroll
    for (int u = 0; u < kFgU; u++) {
      const int i = blk_beg + tid + u * NT;
      ok[u] = i < blk_end;
      const int ii = ok[u] ? i : (blk_beg < a.n ? blk_beg : 0);
      bv[u] = *reinterpret_cast<const double2 *>(b.sb + 2 * (size_t)ii);
      dt[u] = b.sdt[ii];
    }
    __syncthreads();  // tile_bits cleared
#pragma unroll
    for (int u = 0; u < kFgU; u++) {
      const FeWarp w = fe_warp_math<true>(a, bv[u].x, bv[u].y, 1.0, dt[u]);
      ok[u] = ok[u] && w.ok;
      off[u] = ok[u] ? (unsigned)(((size_t)w.yy * W + w.xx) * sizeof(float)) : 0u;  // (cells 0 .. W+1 exist in every image the path accepts)
      dx[u] = w.dx;
      dy[u] = w.dy;
#pragma unroll
      for (int k = 0; k < 3; k++) { r0[u][k] = ok[u] ? w.r0[k] : 0.f; r1[u][k] = ok[u] ? w.r1[k] : 0.f; }
      if (ok[u]) {
        // the (up to four) sort tiles the vote's 2 x 2 cells lie in
        const int t00 = (w.yy / kBinTile) * f.tiles_x + w.xx / kBinTile, t01 = (w.yy / kBinTile) * f.tiles_x + (w.xx + 1) / kBinTile;
        const int t10 = ((w.yy + 1) / kBinTile) * f.tiles_x + w.xx / kBinTile, t11 = ((w.yy + 1) / kBinTile) * f.tiles_x + (w.xx + 1) / kBinTile;
        atomicOr(&sm.tile_bits[t00 >> 5], 1u << (t00 & 31));
        if (t01 != t00) atomicOr(&sm.tile_bits[t01 >> 5], 1u << (t01 & 31));
        if (t10 != t00) atomicOr(&sm.tile_bits[t10 >> 5], 1u << (t10 & 31));
        if (t11 != t10 && t11 != t01) atomicOr(&sm.tile_bits[t11 >> 5], 1u << (t11 & 31));
        if (w.xx <= r || w.xx + 1 >= W - 1 - r || w.yy <= r || w.yy + 1 >= H - 1 - r) {  // votes within r of the border: the mu term's c = G^T 1
          const float c00 = f.cx[w.xx] * f.cy[w.yy], c01 = f.cx[w.xx + 1] * f.cy[w.yy], c10 = f.cx[w.xx] * f.cy[w.yy + 1],
                      c11 = f.cx[w.xx + 1] * f.cy[w.yy + 1];
          const float Ac = (1.f - w.dy) * (c01 - c00) + w.dy * (c11 - c10), Bc = (1.f - w.dx) * (c10 - c00) + w.dx * (c11 - c01);
          if (Ac != 0.f || Bc != 0.f) {
#pragma unroll
            for (int k = 0; k < 3; k++) acc2[k] += (double)w.r0[k] * (double)Ac + (double)w.r1[k] * (double)Bc;
          }
        }
      }
    }
  }
  __syncthreads();
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 1] = wall_clock64();
  // ---- phase 2: wait for the strips of the marked tiles (one polling lane per strip)
  {
    const int ntiles = f.tiles_x * f.tiles_y;
    const unsigned long long t0 = wall_clock64();
    bool give_up = false;
    for (int s0 = 0; s0 < ntiles * kFuseStrips; s0 += NT) {
      const int s = s0 + tid, t = s / kFuseStrips;
      bool need = s < ntiles * kFuseStrips && ((sm.tile_bits[t >> 5] >> (t & 31)) & 1u) && f.nbr_expected[s] > 0;
      while (need) {
        if (__hip_atomic_load(f.tile_done + (size_t)s * kFuseCntStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.seq) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000ull) { give_up = true; break; }
      }
    }
    if (give_up) atomicOr(b.fallback, kFuseIncomplete);
  }
  __syncthreads();
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 2] = wall_clock64();
  // ---- phase 3: the four Jt cells of every event, then the sums
  double acc[3] = {0, 0, 0};
  {
    float c[kFgU][4];
    fg_ld16_sc1(f.jt, f.jt + W, off, c);
#pragma unroll
    for (int u = 0; u < kFgU; u++) {
      const float A = (1.f - dy[u]) * (c[u][1] - c[u][0]) + dy[u] * (c[u][3] - c[u][2]);
      const float B = (1.f - dx[u]) * (c[u][2] - c[u][0]) + dx[u] * (c[u][3] - c[u][1]);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double t = (double)r0[u][k] * (double)A + (double)r1[u][k] * (double)B;
        acc[k] = ok[u] ? acc[k] + t : acc[k];
      }
    }
  }
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 4] = wall_clock64();
  double v[6];
#pragma unroll
  for (int k = 0; k < 3; k++) { v[k] = fg_wave_sum(acc[k]); v[3 + k] = fg_wave_sum(acc2[k]); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) sm.red[wave * 6 + k] = v[k];
  }
  __syncthreads();
  if (tid < 6) {
    double s = 0;
    for (int w = 0; w < NT / 64; w++) s += sm.red[w * 6 + tid];
    if (s != 0.0)
      __hip_atomic_fetch_add(f.gacc + (size_t)(g % kTailShards) * f.gacc_stride + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- last arriver (tail_arrive's protocol, cmx_kernels.hip: sharded tickets, the completing arrivals reset what they completed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int nblocks = f.gather_blocks;
    const int nshards = nblocks < kTailShards ? nblocks : kTailShards;
    const int shard = g % kTailShards;
    const unsigned shard_size = (unsigned)((nblocks - shard + kTailShards - 1) / kTailShards);
    unsigned *cs = f.tail_counters + shard * kTailStride, *ct = f.tail_counters + kTailShards * kTailStride;
    int last = 0;
    if (atomicAdd(cs, 1u) == shard_size - 1u) {
      __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(ct, 1u) == (unsigned)nshards - 1u) {
        __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    sm.is_last = last;
  }
  __syncthreads();
  if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 5] = wall_clock64();
  if (sm.is_last) {
    fg_finalize<NT>(f, b.fallback, sm);
    if (f.trace && tid == 0) f.trace[8 * (size_t)blockIdx.x + 6] = wall_clock64();
  }
}

}  // namespace cmx
