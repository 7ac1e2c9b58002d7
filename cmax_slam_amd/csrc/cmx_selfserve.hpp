// cmx_selfserve.hpp -- the SELF-SERVICE one-launch front-end evaluation (fe_splat_lds_kernel<.., FUSE = 3>; round 6).
//
// One launch of exactly the chunk table's workgroups, nothing behind them.  Every workgroup
//   1. votes its chunk into the LDS window, flushes it, drains its atomics and arrives on the (up to) 25 tile counters around its tile
//      -- the two-launch form's chunk role, unchanged;
//   2. runs the adjoint image pass (cmx_tilepass.hpp) of the tiles it OWNS: tile t belongs to workgroup G - 1 - (t mod G), i.e. to the
//      workgroups at the END of the chunk table first.  The table is ordered longest chunk first, so the owners are the workgroups
//      that finish their votes earliest; with G >= tiles (every dense packet) the G - tiles largest chunks own nothing;
//   3. gathers the gradient sums of ITS OWN events, four per thread and round: the chunk's slice of the tile-ordered streams (read a
//      second time, from L2) is warped again with the Jacobian rows -- the first round BEFORE the wait --, the workgroup waits for the
//      passes of the 3 x 3 tiles its window lies in (tile_done[t] == seq) -- or, if any of its votes took the global path, for every
//      pass of the launch --, reads the four Jt cells of each event (two 8-byte loads; cached loads where a line of Jt belongs to one
//      tile, agent-scope loads otherwise: ss_ld_cells), and adds its six sums to the accumulator rows;
//   4. arrives; the last arriver runs fg_finalize (cmx_fusedgather.hpp).
// What the form removes from the two-launch evaluation: the gather launch's dispatch, the kernel boundary in front of it, and the
// residency pressure of the first one-launch form (FUSE = 2: 713 + 489 workgroups with three roles competing for the CUs).
//
// Waiting.  Every workgroup completes step 1 without waiting for anybody; every wait of steps 2-4 points at step-1 arrivals or at
// passes whose owners wait for step-1 arrivals only.  That is deadlock-free exactly when ALL workgroups of the launch are resident
// at once: the host uses the form only when the table's exact length is known and fits the device (fe_selfserve_capacity), and
// every wait is bounded all the same (another context's kernels can take the CUs): a workgroup that gives up raises
// kFuseIncomplete and the host repeats the evaluation through the separate launches (cmx_frontend.cpp).
//
// Arithmetic: the splat's, the tile pass's and fe_gather_kernel's, operation for operation (reference: local_image_warped_events.cpp
// :94-166 warp + votes + derivative weights, local_focus_funcs.cpp:26-44 variance and its gradient, in the adjoint form of DESIGN.md
// section 4.2); only the grouping of the fp64 partial sums differs (per chunk instead of per 1024-event slice).
#pragma once
#include "cmx_fusedgather.hpp"
#include "cmx_tilepass.hpp"

namespace cmx {

// The four Jt cells of four events: two 8-byte loads per event -- cells (yy, xx), (yy, xx + 1) and the pair one image row below; global
// loads need dword alignment only --, all in flight at once, ONE wait, in ONE asm statement: a register that a pending load is going
// to write must not be visible to the compiler before the wait.
// SC1 = false (production): plain loads, cached by the XCD's L2 and the CU's L1, behind the agent-scope acquire that follows the wait for
// the tiles' stamps (self_serve_tail): a line fetched afterwards holds what its tile's owner released before the stamp.
// SC1 = true (diagnostics): agent-scope loads, served by the memory side every time -- a scattered gather of 2 x 8 bytes per event then
// moves a whole line across the fabric per load (measured: 11-22 us for a chunk's cells against ~1 us).
#define CMX_SS_LD(n, o, r, fl) "global_load_dwordx2 %" #n ", %" #o ", %" #r fl "\n\t"
#define CMX_SS_LD4(fl) CMX_SS_LD(0, 8, 12, fl) CMX_SS_LD(1, 8, 13, fl) CMX_SS_LD(2, 9, 12, fl) CMX_SS_LD(3, 9, 13, fl) CMX_SS_LD(4, 10, 12, fl) \
    CMX_SS_LD(5, 10, 13, fl) CMX_SS_LD(6, 11, 12, fl) CMX_SS_LD(7, 11, 13, fl) "s_waitcnt vmcnt(0)"
template <bool SC1>
__device__ __forceinline__ void ss_ld_cells(const float *row0, const float *row1, const unsigned (&off)[4], unsigned long long (&c)[4][2]) {
#define CMX_SS_OPS4                                                                                                                      \
  : "=&v"(c[0][0]), "=&v"(c[0][1]), "=&v"(c[1][0]), "=&v"(c[1][1]), "=&v"(c[2][0]), "=&v"(c[2][1]), "=&v"(c[3][0]), "=&v"(c[3][1]) \
  : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(row0), "s"(row1)                                                         \
  : "memory"
  if (SC1) asm volatile(CMX_SS_LD4(" sc1") CMX_SS_OPS4);
  else asm volatile(CMX_SS_LD4("") CMX_SS_OPS4);
#undef CMX_SS_OPS4
}
#undef CMX_SS_LD4
#undef CMX_SS_LD
// the two directional differences of Jt at an event's vote cell (bilinear_grad's expressions, cmx_kernels.hip)
__device__ __forceinline__ void ss_cell_grad(const unsigned long long (&c)[2], float dx, float dy, float &A, float &B) {
  const float i00 = __uint_as_float((unsigned)c[0]), i01 = __uint_as_float((unsigned)(c[0] >> 32));
  const float i10 = __uint_as_float((unsigned)c[1]), i11 = __uint_as_float((unsigned)(c[1] >> 32));
  A = (1.f - dy) * (i01 - i00) + dy * (i11 - i10);
  B = (1.f - dx) * (i10 - i00) + dx * (i11 - i01);
}

struct SsSmem {
  FgSmem fg;
  int ok_sh;
};

// the wait of a tile's pass for its inputs: one polling lane, the other waves parked at the barrier (the tile role's wait of the
// two-launch form)
// `early`: thread 0's read of the counter from BEFORE the pass's preamble (operator rows, addresses): an owner is usually late for its
// tile -- the count is complete by then and the poll's round trip to the memory side is not paid a second time behind the preamble
template <int NT>
__device__ __forceinline__ bool ss_wait_tile_inputs(const FusedArgs &f, unsigned *fallback, int t, unsigned expected, unsigned early, int &ok_sh) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    unsigned v = early;
    while (v < expected) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 200000ull) { ok = 0; break; }
      v = __hip_atomic_load(f.nbr_cnt + (size_t)t * kFuseCntStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ok) __hip_atomic_store(f.nbr_cnt + (size_t)t * kFuseCntStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all-zero again for the next launch
    else atomicOr(fallback, kFuseIncomplete);
    ok_sh = ok;
  }
  __syncthreads();
  return ok_sh != 0;
}

// the state of four events of a thread between their warp and their Jt cells: nine registers per event
struct SsRound {
  unsigned off[4];  // byte offset of cell (yy, xx) in Jt
  float dx[4], dy[4], r0[4][3], r1[4][3];
};

// the tile-ordered streams of events i0, i0 + NT, i0 + 2 NT, i0 + 3 NT (< end): bearing (x, y) and dt, 24 bytes per event
struct SsStreams {
  double2 bv[4];
  double dt[4];
  bool ok[4];
};
template <int NT>
__device__ __forceinline__ void ss_load_streams(const BinnedEvents &b, int i0, int end, SsStreams &t) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + q * NT;
    t.ok[q] = i < end;
    const int ii = t.ok[q] ? i : 0;
    t.bv[q] = *reinterpret_cast<const double2 *>(b.sb + 2 * (size_t)ii);
    t.dt[q] = b.sdt[ii];
  }
}
// fp64 warp with the Jacobian rows and the border term of four loaded events: nothing of this depends on the launch's votes.  An
// event that does not count keeps zero rows: its terms in ss_consume are exact zeros whatever cell 0 holds.
__device__ __forceinline__ void ss_warp(const FeSplatArgs &a, const FusedArgs &f, const SsStreams &t, SsRound &s, double (&acc2)[3]) {
  const int W = a.W, H = a.H, r = kTpR;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const FeWarp w = fe_warp_math<true>(a, t.bv[q].x, t.bv[q].y, 1.0, t.dt[q]);
    const bool ok = t.ok[q] && w.ok;
    s.off[q] = ok ? (unsigned)(((size_t)w.yy * W + w.xx) * sizeof(float)) : 0u;  // (cells 0 .. W+1 exist in every image the path accepts)
    s.dx[q] = w.dx;
    s.dy[q] = w.dy;
#pragma unroll
    for (int k = 0; k < 3; k++) { s.r0[q][k] = ok ? w.r0[k] : 0.f; s.r1[q][k] = ok ? w.r1[k] : 0.f; }
    if (ok && (w.xx <= r || w.xx + 1 >= W - 1 - r || w.yy <= r || w.yy + 1 >= H - 1 - r)) {  // votes within r of the border: the mu term's c = G^T 1
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
template <int NT>
__device__ __forceinline__ void ss_prewarp(const FeSplatArgs &a, const BinnedEvents &b, const FusedArgs &f, int i0, int end, SsRound &s, double (&acc2)[3]) {
  SsStreams t;
  ss_load_streams<NT>(b, i0, end, t);
  ss_warp(a, f, t, s, acc2);
}
// the four Jt cells of a round's events and their terms of the three sums
__device__ __forceinline__ void ss_consume(const float *row0, const float *row1, bool plain, const SsRound &s, double (&acc)[3]) {
  unsigned long long cell[4][2];
  if (plain) ss_ld_cells<false>(row0, row1, s.off, cell);
  else ss_ld_cells<true>(row0, row1, s.off, cell);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float A, B;
    ss_cell_grad(cell[q], s.dx[q], s.dy[q], A, B);
#pragma unroll
    for (int k = 0; k < 3; k++) acc[k] += (double)s.r0[q][k] * (double)A + (double)s.r1[q][k] * (double)B;
  }
}

// Steps 2-4 of a self-service workgroup.  `c`: its chunk (has_chunk = false: a workgroup beyond the table's length -- it owns tiles
// and arrives, nothing else); `global_votes`: any of its votes took the global-atomic path; lds: the vote window, free again.
template <int NT>
__device__ __forceinline__ void self_serve_tail(const FeSplatArgs &a, const BinnedEvents &b, const FusedArgs &f, const Chunk &c, bool has_chunk,
                                                bool global_votes, unsigned char *lds, SsSmem &sm) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int W = a.W;
  const int G = (int)gridDim.x, ntiles = f.tiles_x * f.tiles_y;
  unsigned long long *tr = f.trace ? f.trace + 8 * (size_t)blockIdx.x : nullptr;
  const int beg = has_chunk ? c.beg : 0, end = has_chunk ? c.end : 0;
  double acc[3] = {0, 0, 0}, acc2[3] = {0, 0, 0};
  SsRound s0;
  // ---- step 2: the passes of the tiles this workgroup owns
  FusedArgs fq = f;
  fq.trace = nullptr;  // (the pass's own stamps belong to the two-launch form's layout)
  for (int t = G - 1 - (int)blockIdx.x; t < ntiles; t += G) {
    const unsigned expected = (unsigned)f.nbr_expected[t];
    if (expected == 0u) continue;  // no vote can reach this tile: B = Jt = 0 there, zero moments (rows cleared at sort time)
    bool ran = false;
    unsigned early = 0u;
    if (tid == 0) early = __hip_atomic_load(f.nbr_cnt + (size_t)t * kFuseCntStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fused_tile_pass<NT, true>(fq, a.planes, W, a.H, t, lds, [&]() -> bool {
      ran = ss_wait_tile_inputs<NT>(f, b.fallback, t, expected, early, sm.ok_sh);
      if (tr && tid == 0) tr[2] = wall_clock64();
      return ran;
    });
    // publish: every wave's write-through stores have left (the producer side of the tail finalize's protocol, cmx_kernels.hip), then
    // the tile's stamp and the launch's count of passes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(f.tile_done + (size_t)t * kFuseCntStride, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(f.tiles_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tr) tr[4] = wall_clock64();
    }
  }
  // ---- step 3a (nothing here depends on this launch's votes): the thread's first four events -- streams, warp, Jacobian rows, border
  // term -- while the neighbours' passes finish.  Measured alternatives (profiles/r06_selfserve.txt): in FRONT of the own pass (the
  // warp of 2048 events is ~1.5 us of fp64 issue on the workgroup's CU, twice that with two workgroups on it: it delays the pass, which
  // every neighbour waits for, and the state kept across the pass spills); eight events per thread here (spills: every scratch reload
  // behind the wait is a memory round trip).
  ss_prewarp<NT>(a, b, f, beg + tid, end, s0, acc2);
  // chunks of more than 4 NT events (workgroup-uniform): the second round's streams are requested in front of the wait as well
  const bool two = end - beg > 4 * NT;
  SsStreams t1;
  if (two) ss_load_streams<NT>(b, beg + 4 * NT + tid, end, t1);
  if (tr && tid == 0) tr[5] = wall_clock64();
  // ---- step 3b: wait for the passes this chunk's vote cells lie in: the 3 x 3 tiles around its own (window = tile + 16 px); with
  // votes on the global path (or no window at all) for every pass of the launch
  if (has_chunk && end > beg) {
    if (tid < 16) {
      const unsigned long long t0 = wall_clock64();
      bool need = false, all = false;
      const unsigned *word = nullptr;
      unsigned want = 0u;
      if (tid < 9 && c.tile >= 0) {
        const int tx = c.tile % f.tiles_x + (tid % 3 - 1), ty = c.tile / f.tiles_x + (tid / 3 - 1);
        if (tx >= 0 && tx < f.tiles_x && ty >= 0 && ty < f.tiles_y && f.nbr_expected[ty * f.tiles_x + tx] > 0) {
          need = true;
          word = f.tile_done + (size_t)(ty * f.tiles_x + tx) * kFuseCntStride;
          want = f.seq;
        }
      } else if (tid == 9 && (global_votes || c.tile < 0)) {
        need = all = true;
        word = f.tiles_done;
        want = (unsigned)*f.n_active;
      }
      bool give_up = false;
      while (need) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (all ? v >= want : v == want) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000ull) { give_up = true; break; }
      }
      if (give_up) atomicOr(b.fallback, kFuseIncomplete);
    }
    __syncthreads();
    // agent-scope ACQUIRE behind the stamps (every wave; buffer_inv sc1: the CU's L1 and the XCD's L2 drop what they hold of memory other
    // XCDs write): the cell loads below are plain, cached loads.  This is what the form costs: 30 -> 48 us per launch.  WITHOUT it the
    // launch is right on a GPU the context has to itself (both caches are invalidated when a launch starts, and a line of Jt is loaded
    // only behind its tile's stamp) and WRONG about once in 2000 evaluations -- a few stale cells from the previous evaluation, in
    // the XCD's L2: an L1-only invalidate does not cure it -- when three contexts share the GPU (profiles/r06_selfserve.txt section 6).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (tr && tid == 0) tr[6] = wall_clock64();
  // ---- step 3c: the four Jt cells of every event, the sums
  const float *row0 = f.jt, *row1 = f.jt + W;
  const bool plain = !(f.debug & 256);  // (diagnostics: agent-scope loads instead, see ss_ld_cells)
  ss_consume(row0, row1, plain, s0, acc);
  if (tr && tid == 0 && (f.debug & 64)) tr[2] = wall_clock64();
  if (two) {
    ss_warp(a, f, t1, s0, acc2);
    ss_consume(row0, row1, plain, s0, acc);
  }
  for (int i0 = beg + 8 * NT + tid; i0 < end; i0 += 4 * NT) {  // chunks of more than 8 NT events (packets above ~1M events): further rounds
    ss_prewarp<NT>(a, b, f, i0, end, s0, acc2);
    ss_consume(row0, row1, plain, s0, acc);
  }
  if (tr && tid == 0 && (f.debug & 64)) tr[4] = wall_clock64();
  FgSmem &fs = sm.fg;
  double v[6];
#pragma unroll
  for (int k = 0; k < 3; k++) v[k] = fg_wave_sum(acc[k]);
  const bool any2 = __any(acc2[0] != 0.0 || acc2[1] != 0.0 || acc2[2] != 0.0);
#pragma unroll
  for (int k = 0; k < 3; k++) v[3 + k] = any2 ? fg_wave_sum(acc2[k]) : 0.0;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) fs.red[wave * 6 + k] = v[k];
  }
  __syncthreads();
  if (tid < 6) {
    double s = 0;
    for (int w = 0; w < NT / 64; w++) s += fs.red[w * 6 + tid];
    if (s != 0.0)
      __hip_atomic_fetch_add(f.gacc + (size_t)(blockIdx.x % kTailShards) * f.gacc_stride + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tr && tid == 0) tr[7] = wall_clock64();
  // ---- step 4: last arriver (tail_arrive's protocol, cmx_kernels.hip: sharded tickets, the completing arrivals reset what they completed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int nshards = G < kTailShards ? G : kTailShards;
    const int shard = (int)blockIdx.x % kTailShards;
    const unsigned shard_size = (unsigned)((G - shard + kTailShards - 1) / kTailShards);
    unsigned *cs = f.tail_counters + shard * kTailStride, *ct = f.tail_counters + kTailShards * kTailStride;
    int last = 0;
    if (atomicAdd(cs, 1u) == shard_size - 1u) {
      __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(ct, 1u) == (unsigned)nshards - 1u) {
        __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    fs.is_last = last;
  }
  __syncthreads();
  if (fs.is_last) fg_finalize<NT>(f, b.fallback, fs);
}

}  // namespace cmx
