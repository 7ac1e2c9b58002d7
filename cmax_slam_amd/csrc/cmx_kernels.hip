// cmx_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for cmax_slam's event-warping hot path.
//
//   fe_splat        K1  front-end warp + bilinear splat      (reference local_image_warped_events.cpp:94-169)
//   be_pose_table   K0  per-batch spline pose + Jacobian     (reference event_pano_warper.cpp:239-256, so3_spline.h:218-274)
//   be_splat        K2  back-end rotate + equirect + splat   (reference event_pano_warper.cpp:262-335,
//                                                             equirectangular_camera.h:18-45)
//   image_moments   K3+K5+K6  compose I = IL + alpha*IGp, separable Gaussian (REFLECT_101), moment reduction
//                                                            (reference event_pano_warper.cpp:199-230,
//                                                             local_focus_funcs.cpp:9-44, global_focus_funcs.cpp:11-47)
//   alpha_*         K4  event-density ratio alpha            (reference event_pano_warper.cpp:134-165)
//   image_adjoint   fused blur + moments + G^T (adjoint gradient), fe/be_gather: gradient by gathering over the events
//   reduce/finalize     partial moments -> contrast, gradient (fp64)
//
// Numerics: geometry in fp64 exactly as the reference, weights/accumulators fp32.  This file is compiled with
// -ffp-contract=off so the fp64 warp, the fp32 weights and the fp32 blur are bit-identical to the CPU path for
// identical inputs; the only reordering is the fp32 atomic accumulation.
#include "../../include/cmax_hip.h"
#include "cmx_internal.hpp"
#include "cmx_warp.hpp"

// launch with optional kernel-exact timing events (null, null = plain launch)
#define CMX_LAUNCH(kernel, grid, block, lds, stream, t0, t1, ...)                                            \
  do {                                                                                                      \
    if ((t0) || (t1)) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, t0, t1, 0, __VA_ARGS__);      \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
  } while (0)
namespace cmx {

// ---------------------------------------------------------------------------------------------- K1
template <bool DERIV>
__global__ __launch_bounds__(256) void fe_splat_kernel(FeSplatArgs a) {
  const int W = a.W;
  const size_t np = (size_t)W * a.H;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const FeWarp w = fe_warp_event<DERIV>(a, i);
    if (w.ok) {
      const float dx = w.dx, dy = w.dy;
      float *q = a.planes + (size_t)w.yy * W + w.xx;
      atomic_add_f32(q, (1.f - dx) * (1.f - dy));
      atomic_add_f32(q + 1, dx * (1.f - dy));
      atomic_add_f32(q + W, (1.f - dx) * dy);
      atomic_add_f32(q + W + 1, dx * dy);
      if (DERIV) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float r0 = w.r0[k], r1 = w.r1[k];
          float *d = q + (size_t)(1 + k) * np;
          atomic_add_f32(d, r0 * (-(1.f - dy)) + r1 * (-(1.f - dx)));
          atomic_add_f32(d + 1, r0 * (1.f - dy) + r1 * (-dx));
          atomic_add_f32(d + W, r0 * (-dy) + r1 * (1.f - dx));
          atomic_add_f32(d + W + 1, r0 * dy + r1 * dx);
        }
      }
    }
  }
}

static int splat_grid(int n) {
  int blocks = (n + 255) / 256;
  const int cap = 256 * 8;  // 256 CUs x 8 blocks; grid-stride beyond
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

void launch_fe_splat(const FeSplatArgs &a, bool deriv, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (a.n <= 0) return;
  if (deriv) CMX_LAUNCH(fe_splat_kernel<true>, dim3(splat_grid(a.n)), dim3(256), 0, s, t0, t1, a);
  else CMX_LAUNCH(fe_splat_kernel<false>, dim3(splat_grid(a.n)), dim3(256), 0, s, t0, t1, a);
}

// ---------------------------------------------------------------------------------------------- K0
template <int N, bool WANT_J>
__global__ __launch_bounds__(64) void be_pose_table_kernel(const SplineArgs sp, const long long *batch_t, int nb,
                                                           PoseR *outR, PoseEntry *out) {  // sp by value: 2.2 KB of
  // kernel arguments instead of a host-to-device copy (a 5 us copy kernel + a boundary) in front of every evaluation
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nb) return;
  Mat3 R, J[N];
  int idx;
  spline_eval<N, WANT_J>(sp, batch_t[b], R, J, idx);
  PoseEntry &o = out[b];
#pragma unroll
  for (int i = 0; i < 9; i++) outR[b].R[i] = R.m[i];
  o.idx_cp_beg = idx;
  if (WANT_J) {
    // 3 x 3N fp32, block k at columns 3k..3k+2  (Trajectory::evaluate copies d_val_d_knot[k] as float)
#pragma unroll
    for (int k = 0; k < N; k++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) o.Jcp[r * (3 * N) + 3 * k + c] = (float)J[k].m[r * 3 + c];
  }
}

// the same table from a spline whose per-pair constants the host has already evaluated (K <= kMaxKnotsPre)
// A thread's work hangs on three dependent memory round trips of ~1.2 us each -- kernel arguments, batch_t[b], then the
// constants of the segment batch_t[b] falls into (PMC: 61 % of the wave cycles are waits, profiles/r02_pose_kernarg.txt).
// The constants do not depend on b: the workgroup copies all of them to LDS while batch_t[b] is in flight.
constexpr int kPoseThreads = 256;
template <int N, bool WANT_J>
__global__ __launch_bounds__(kPoseThreads) void be_pose_table_pre_kernel(const SplineArgsPre sp, const long long *batch_t, int nb,
                                                                         PoseR *outR, PoseEntry *out) {
  __shared__ Quat sh_knots[kMaxKnotsPre];
  __shared__ PairConsts sh_pair[kMaxKnotsPre - 1];
  const int b = blockIdx.x * kPoseThreads + threadIdx.x;
  const bool live = b < nb;
  const long long t_ns = batch_t[live ? b : nb - 1];
  {
    const double *gk = reinterpret_cast<const double *>(sp.knots), *gp = reinterpret_cast<const double *>(sp.pair);
    double *lk = reinterpret_cast<double *>(sh_knots), *lp = reinterpret_cast<double *>(sh_pair);
    const int nk = sp.K * 4, npd = (sp.K - 1) * (int)(sizeof(PairConsts) / sizeof(double));
    for (int i = threadIdx.x; i < nk; i += kPoseThreads) lk[i] = gk[i];
    for (int i = threadIdx.x; i < npd; i += kPoseThreads) lp[i] = gp[i];
  }
  __syncthreads();
  if (!live) return;
  Mat3 R, J[N];
  int idx;
  spline_eval_pre<N, WANT_J>(sp, sh_knots, sh_pair, t_ns, R, J, idx);
  PoseEntry &o = out[b];
#pragma unroll
  for (int i = 0; i < 9; i++) outR[b].R[i] = R.m[i];
  o.idx_cp_beg = idx;
  if (WANT_J) {
#pragma unroll
    for (int k = 0; k < N; k++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) o.Jcp[r * (3 * N) + 3 * k + c] = (float)J[k].m[r * 3 + c];
  }
}

void launch_be_pose_table(const SplineArgs &spline, const long long *d_batch_t, int nb, int order, bool want_j,
                          PoseR *outR, PoseEntry *out, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (nb <= 0) return;
  const dim3 g((nb + 63) / 64), b(64);
  if (spline.K <= kMaxKnotsPre) {
    SplineArgsPre pre;
    spline_precompute(spline, pre);
    const dim3 gp((nb + kPoseThreads - 1) / kPoseThreads), bp(kPoseThreads);
    if (order == 2) {
      if (want_j) CMX_LAUNCH((be_pose_table_pre_kernel<2, true>), gp, bp, 0, s, t0, t1, pre, d_batch_t, nb, outR, out);
      else CMX_LAUNCH((be_pose_table_pre_kernel<2, false>), gp, bp, 0, s, t0, t1, pre, d_batch_t, nb, outR, out);
    } else {
      if (want_j) CMX_LAUNCH((be_pose_table_pre_kernel<4, true>), gp, bp, 0, s, t0, t1, pre, d_batch_t, nb, outR, out);
      else CMX_LAUNCH((be_pose_table_pre_kernel<4, false>), gp, bp, 0, s, t0, t1, pre, d_batch_t, nb, outR, out);
    }
    return;
  }
  if (order == 2) {
    if (want_j) CMX_LAUNCH((be_pose_table_kernel<2, true>), g, b, 0, s, t0, t1, spline, d_batch_t, nb, outR, out);
    else CMX_LAUNCH((be_pose_table_kernel<2, false>), g, b, 0, s, t0, t1, spline, d_batch_t, nb, outR, out);
  } else {
    if (want_j) CMX_LAUNCH((be_pose_table_kernel<4, true>), g, b, 0, s, t0, t1, spline, d_batch_t, nb, outR, out);
    else CMX_LAUNCH((be_pose_table_kernel<4, false>), g, b, 0, s, t0, t1, spline, d_batch_t, nb, outR, out);
  }
}

// ---------------------------------------------------------------------------------------------- K2
template <int N, bool DERIV>
__global__ __launch_bounds__(256) void be_splat_kernel(BeSplatArgs a) {
  const int Wp = a.Wp;
  const size_t np = (size_t)Wp * a.Hp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const BeWarp w = be_warp_event<DERIV ? 1 : 0>(a, i);
    if (w.ok) {
      const float dx = w.dx, dy = w.dy;
      const size_t off = (size_t)w.yy * Wp + w.xx;
      float *q = a.planes + (w.is_old ? 0 : np) + off;
      atomic_add_f32(q, (1.f - dx) * (1.f - dy));
      atomic_add_f32(q + 1, dx * (1.f - dy));
      atomic_add_f32(q + Wp, (1.f - dx) * dy);
      atomic_add_f32(q + Wp + 1, dx * dy);
      if (DERIV) {
        const PoseEntry &pe = a.poses[w.batch];
        const int jbase = 3 * (pe.idx_cp_beg - a.num_fixed);
#pragma unroll
        for (int c = 0; c < 3 * N; c++) {
          const int j = jbase + c;
          if (j >= 0) {
            // jac = dpm_ddrot(2x3) * ddrot_ddrot_cp(3x3N): fp64 accumulation, fp32 result
            const double j0 = (double)pe.Jcp[c], j1 = (double)pe.Jcp[3 * N + c], j2 = (double)pe.Jcp[6 * N + c];
            const float r0 = (float)((double)w.m[0] * j0 + (double)w.m[1] * j1 + (double)w.m[2] * j2);
            const float r1 = (float)((double)w.m[3] * j0 + (double)w.m[4] * j1 + (double)w.m[5] * j2);
            float *d = a.planes + (size_t)(2 + j) * np + off;
            atomic_add_f32(d, r0 * (-(1.f - dy)) + r1 * (-(1.f - dx)));
            atomic_add_f32(d + 1, r0 * (1.f - dy) + r1 * (-dx));
            atomic_add_f32(d + Wp, r0 * (-dy) + r1 * (1.f - dx));
            atomic_add_f32(d + Wp + 1, r0 * dy + r1 * dx);
          }
        }
      }
    }
  }
}

void launch_be_splat(const BeSplatArgs &a, bool deriv, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (a.n <= 0) return;
  const dim3 g(splat_grid(a.n)), b(256);
  if (a.order == 2) {
    if (deriv) CMX_LAUNCH((be_splat_kernel<2, true>), g, b, 0, s, t0, t1, a);
    else CMX_LAUNCH((be_splat_kernel<2, false>), g, b, 0, s, t0, t1, a);
  } else {
    if (deriv) CMX_LAUNCH((be_splat_kernel<4, true>), g, b, 0, s, t0, t1, a);
    else CMX_LAUNCH((be_splat_kernel<4, false>), g, b, 0, s, t0, t1, a);
  }
}

// ---------------------------------------------------------------------------------------------- K3+K5+K6
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// wave64 sum of a double with DPP moves (no LDS crossbar): inclusive row_shr scan inside each row of 16 lanes, then
// row_bcast15 / row_bcast31 carry the row totals upwards; lane 63 holds the total, read back as a wave-uniform value
// (valid in every lane).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);  // lanes without a source add 0
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row = row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// block-wide sum of a per-thread double; result valid in thread 0.  red: 4 doubles of LDS per call site.
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// two block-wide sums behind one pair of barriers (wave order as in block_sum); red: 2 * nwaves doubles
__device__ __forceinline__ void block_sum2(double v0, double v1, double *red, int nwaves, double &t0, double &t1) {
  v0 = wave_sum(v0);
  v1 = wave_sum(v1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { red[2 * wave] = v0; red[2 * wave + 1] = v1; }
  __syncthreads();
  t0 = 0;
  t1 = 0;
  for (int w = 0; w < nwaves; w++) { t0 += red[2 * w]; t1 += red[2 * w + 1]; }
}

// Tile occupancy (ImgArgs::flags_*): is there anything non-zero within `reach` pixels of tile (tx, ty)?
__device__ __forceinline__ bool tile_active(const ImgArgs &a, int tx, int ty, int reach, int TX, int TY) {
  if (!a.flags_cur) return true;
  const int nx = (reach + TX - 1) / TX, ny = (reach + TY - 1) / TY;
  bool any = false;
  for (int dy = -ny; dy <= ny; dy++)
    for (int dx = -nx; dx <= nx; dx++) {
      const int x = tx + dx, y = ty + dy;
      if (x < 0 || y < 0 || x >= a.tiles_x || y >= a.tiles_y) continue;  // REFLECT_101 mirrors pixels of the same tiles
      const int t = y * a.tiles_x + x;
      any = any || a.flags_cur[t] != 0 || (a.igp && a.flags_igp && a.flags_igp[t] != 0);
    }
  return any;
}

__global__ __launch_bounds__(256) void tile_flags_kernel(const float *plane, int W, int H, int tiles_x, unsigned char *flags) {
  const size_t n = (size_t)W * H;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (plane[i] != 0.f) {
      const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
      flags[(y / kTileY) * tiles_x + x / kTileX] = 1;
    }
}
// occupancy of TWO summed planes (IL_old, IL_new after a whole-plane all-reduce): one workgroup per tile writes its flag,
// 0 or 1 -- no memset in front, one launch
__global__ __launch_bounds__(256) void tile_flags_pair_kernel(const float *a, const float *b, int W, int H, int tiles_x, unsigned char *flags) {
  const int tile = blockIdx.x, x0 = (tile % tiles_x) * kTileX, y0 = (tile / tiles_x) * kTileY;
  bool any = false;
  for (int q = threadIdx.x; q < kTileX * kTileY; q += 256) {
    const int x = x0 + (q % kTileX), y = y0 + (q / kTileX);
    if (x < W && y < H) {
      const size_t o = (size_t)y * W + x;
      any = any || a[o] != 0.f || b[o] != 0.f;
    }
  }
  const int r = __syncthreads_or(any ? 1 : 0);
  if (threadIdx.x == 0) flags[tile] = r ? 1 : 0;
}
void launch_tile_flags_pair(const float *a, const float *b, int W, int H, unsigned char *flags, hipStream_t s) {
  const int tiles_x = (W + kTileX - 1) / kTileX, tiles_y = (H + kTileY - 1) / kTileY;
  hipLaunchKernelGGL(tile_flags_pair_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, s, a, b, W, H, tiles_x, flags);
}
void launch_tile_flags(const float *plane, int W, int H, unsigned char *flags, hipStream_t s) {
  const int tiles_x = (W + kTileX - 1) / kTileX;
  hipLaunchKernelGGL(tile_flags_kernel, dim3(1024), dim3(256), 0, s, plane, W, H, tiles_x, flags);
}

// One thread per tile: classify it active / dirty from the occupancy flags of its neighbourhood, un-flag the partner's
// dirty tiles, compact the listed tiles (wave ballot -> workgroup offsets -> one atomic per workgroup on the list
// counter).  The order of the list does not matter (tiles are independent).  count[0] is this pass's counter, count[1]
// the next pass's: it is zeroed here, so no memset sits between two passes (the host alternates the pair).
__global__ __launch_bounds__(1024) void tile_list_kernel(ImgArgs a, int reach, unsigned *list, unsigned *count, unsigned *next_count) {
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned block_base;
  const int ntiles = a.tiles_x * a.tiles_y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x * 1024 + tid;
  if (t == 0) *next_count = 0u;
  const int nx = (reach + kTileX - 1) / kTileX, ny = (reach + kTileY - 1) / kTileY;
  bool listed = false;
  unsigned entry = 0;
  if (t < ntiles) {
    const int tx = t % a.tiles_x, ty = t / a.tiles_x;
    const bool dirty = a.zero_ptr && (!a.flags_other || a.flags_other[t] != 0);
    bool active = false;
    for (int dy = -ny; dy <= ny; dy++)
      for (int dx = -nx; dx <= nx; dx++) {
        const int x = tx + dx, y = ty + dy;
        if (x >= 0 && y >= 0 && x < a.tiles_x && y < a.tiles_y) {
          const int q = y * a.tiles_x + x;
          active = active || a.flags_cur[q] != 0 || (a.igp && a.flags_igp && a.flags_igp[q] != 0);
        }
      }
    if (dirty && a.flags_other) a.flags_other[t] = 0;
    listed = active || dirty;
    entry = (unsigned)t | (active ? 0x80000000u : 0u) | (dirty ? 0x40000000u : 0u);
  }
  const unsigned long long m = __ballot(listed);
  const unsigned before = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = (unsigned)__popcll(m);
  __syncthreads();
  if (tid == 0) {
    unsigned tot = 0;
    for (int w = 0; w < 16; w++) tot += wave_tot[w];
    block_base = tot ? atomicAdd(count, tot) : 0u;
  }
  __syncthreads();
  unsigned off = block_base;
  for (int w = 0; w < wave; w++) off += wave_tot[w];
  if (listed) list[off + before] = entry;
}
// The same list in TILE ORDER, built by one workgroup (no atomics): the order of the list is the order in which finalize
// sums the tiles' moment rows, so with it a result does not depend on which workgroup reserved its slots first.  Used by
// sharded evaluations: the ranks' replicated optimiser drivers must see bit-identical numbers or they would stop taking
// the same decisions (and issuing the same collectives); ranks share flags and history, hence this list.
// Tile-ordered list in two launches: every tile's entry (0 = not listed) by one thread per tile across the GPU, then ONE
// workgroup compacts the dense array in order -- coalesced 16-byte reads, one scan per 16 K tiles.  (One workgroup doing both
// took 34 us at 4096x2048: 16 K tiles x 18 scattered byte loads through a single CU's address path; 46 us with all of a
// thread's loads in one round.)
__global__ __launch_bounds__(1024) void tile_mark_kernel(ImgArgs a, int reach, unsigned *dense) {
  const int ntiles = a.tiles_x * a.tiles_y;
  const int t = blockIdx.x * 1024 + threadIdx.x;
  if (t >= ntiles) return;
  const int nx = (reach + kTileX - 1) / kTileX, ny = (reach + kTileY - 1) / kTileY;
  const int tx = t % a.tiles_x, ty = t / a.tiles_x;
  const bool dirty = a.zero_ptr && (!a.flags_other || a.flags_other[t] != 0);
  bool active = false;
  for (int dy = -ny; dy <= ny; dy++)
    for (int dx = -nx; dx <= nx; dx++) {
      const int x = tx + dx, y = ty + dy;
      if (x >= 0 && y >= 0 && x < a.tiles_x && y < a.tiles_y) {
        const int q = y * a.tiles_x + x;
        active = active || a.flags_cur[q] != 0 || (a.igp && a.flags_igp && a.flags_igp[q] != 0);
      }
    }
  if (dirty && a.flags_other) a.flags_other[t] = 0;
  // bit 29 marks "listed" (tile 0 has an all-zero index); the compaction clears it again
  dense[t] = (active || dirty) ? ((unsigned)t | (active ? 0x80000000u : 0u) | (dirty ? 0x40000000u : 0u) | 0x20000000u) : 0u;
}
__global__ __launch_bounds__(1024) void tile_list_ordered_kernel(const unsigned *dense, int ntiles, unsigned *list, unsigned *count,
                                                                 unsigned *next_count) {
  constexpr int kPer = 16;
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned base_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { *next_count = 0u; base_sh = 0u; }
  __syncthreads();
  for (int t0 = 0; t0 < ntiles; t0 += 1024 * kPer) {
    unsigned entry[kPer];
    unsigned mine = 0;
    const int first = t0 + tid * kPer;
    if (first + kPer <= ntiles && (ntiles & 3) == 0) {  // (the scratch array starts 16-byte aligned: see launch_tile_list)
#pragma unroll
      for (int q = 0; q < kPer / 4; q++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(dense + first + 4 * q);
        entry[4 * q] = v.x; entry[4 * q + 1] = v.y; entry[4 * q + 2] = v.z; entry[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kPer; k++) entry[k] = first + k < ntiles ? dense[first + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) mine += entry[k] != 0u;
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned off = base_sh + incl - mine, tot = 0;
    for (int w = 0; w < 16; w++) {
      if (w < wave) off += wave_tot[w];
      tot += wave_tot[w];
    }
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (entry[k]) list[off++] = entry[k] & ~0x20000000u;
    __syncthreads();
    if (tid == 0) base_sh += tot;
    __syncthreads();
  }
  if (tid == 0) *count = base_sh;
}
void launch_tile_list(const ImgArgs &a, int reach, unsigned *list, unsigned *count, unsigned *next_count, bool ordered, hipStream_t s) {
  const int ntiles = a.tiles_x * a.tiles_y;
  if (ordered) {  // `list` has room for 2 x ntiles entries (rounded up to 4): the second half is the dense scratch array
    unsigned *dense = list + ((ntiles + 3) & ~3);
    hipLaunchKernelGGL(tile_mark_kernel, dim3((ntiles + 1023) / 1024), dim3(1024), 0, s, a, reach, dense);
    hipLaunchKernelGGL(tile_list_ordered_kernel, dim3(1), dim3(1024), 0, s, dense, ntiles, list, count, next_count);
  }
  else hipLaunchKernelGGL(tile_list_kernel, dim3((ntiles + 1023) / 1024), dim3(1024), 0, s, a, reach, list, count, next_count);
}

// contrast / gradient from the moments (fp64):
//   variance:     contrast = (sqrt(max(E[I^2]-mu^2,0)))^2 ; grad_k = 2*(E[I D_k] - mu*E[D_k])
//   mean square:  contrast = E[I^2]                        ; grad_k = 2*E[I D_k]
// One workgroup of NT threads (16 waves as its own kernel; 4 waves when it runs as the tail of the evaluation's last
// kernel, see tail_arrive).  Two groups of waves reduce the two image moments; then every wave reduces whole
// parameters on its own (wave sums over coalesced rows of the [column][block] partial tables): no workgroup barriers in
// the per-parameter loop.  The partial tables are read with agent-scope (sc1) loads: in tail mode other workgroups of
// the SAME launch wrote them (write-through, st_sc1), and a plain load could be served from a stale L1 / L2 line.
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v) {  // write-through store: visible to every XCD without a release fence
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
struct alignas(16) FinSmem {  // a multiple of 16 bytes: static LDS in front of a kernel's dynamic region must not misalign it
  double sh[2];
  double shp[16];
  double outv[2 + 3 * kMaxKnots];  // results are staged here and written to the mapped host buffer by ONE wave,
                                   // contiguously: scattered lane writes over PCIe cost ~0.5 us each
  double cols[2 * 3 * kMaxKnots];  // per-column sums of the gather partial table: S1 (gP) then S2 (gP)
  double colw[16][8];              // per-wave partial column sums (many-rows form)
  double shfall;
  unsigned long long shchk;
  int is_last;
  int pad[3];
  // device-driven solve: the FR-CG machine and its vectors while the finalize advances them (ChainArgs)
  ChainMachine csm;
};
static_assert(sizeof(FinSmem) % 16 == 0, "FinSmem must keep the dynamic LDS base 16-byte aligned");

// the two image moments of a self-gating slot from their 8 accumulator rows, always in this order
__device__ __forceinline__ void chain_moment_sums(const double *macc, double &s0, double &s1) {
  double a0 = 0, a1 = 0;
#pragma unroll
  for (int q = 0; q < kTailShards; q++) {
    a0 += macc[(size_t)q * 16];
    a1 += macc[(size_t)q * 16 + 1];
  }
  s0 = a0;
  s1 = a1;
}

// One step of the device-driven solve's machine inside a finalize (thread 0): feed it what this evaluation produced -- f =
// -contrast after a cost evaluation, df = -gradient after a gradient pass, exactly what the host-driven loop feeds it
// (cmx_solver.cpp: contrast_fdf) -- and publish the next request: the evaluation point for the next slot's kernels, the gate of
// a flag-gated gradient pass, the end of the solve.  outv[nout ..]: need / phase / done (| 2: disagreement) / next point.
__device__ __forceinline__ void chain_step(const FinalizeArgs &a, ChainMachine &s, double *outv, int nout) {
  constexpr int n = kChainMaxN;
  int need = 0, disagree = 0;
  if (a.chain.stage == 0) {
    need = sm_cost(s, -outv[0]) ? 1 : 0;
  } else if (a.chain.stage == 1) {
    double g[n];
    for (int k = 0; k < n; k++) g[k] = -outv[2 + k];
    sm_grad(s, g);
  } else {  // self-gating slot: the launch's workgroups decided from the same contrast whether to compute the gradient
    const bool have_grad = a.gP > 0;
    // (the test the workgroups evaluated: the published copy; first slot of a warm start: "always")
    const double cur_thr = a.chain.sm_src ? 0.0 : a.chain.gate_cur->thr;
    const int cur_mode = a.chain.sm_src ? 4 : a.chain.gate_cur->mode;
    if ((gate_condition(outv[0], cur_thr, cur_mode) != 0) != have_grad || cur_mode != s.gate_mode ||
        __double_as_longlong(cur_thr) != __double_as_longlong(s.gate_thr)) {
      disagree = 1;  // (cannot happen: same number, same expression) -- the machine is left where it was, the host takes over
    } else {
      need = sm_cost(s, -outv[0]) ? 1 : 0;
      if (need != (have_grad ? 1 : 0)) {
        disagree = 1;
      } else if (need) {
        double g[n];
        for (int k = 0; k < n; k++) g[k] = -outv[2 + k];
        sm_grad(s, g);
      }
    }
    if (disagree) __hip_atomic_store(a.chain.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int done = (sm_done(s) || disagree) ? 1 : 0;
  const bool moved = a.chain.stage != 0 || !need;  // the machine has gone on to its next request
  if (a.gate_out) *a.gate_out = need;              // read by the gradient pass queued behind this launch
  if (a.chain.gate_next) {                         // self-gating slots: the NEXT slot's test, in the words this launch does not read
    a.chain.gate_next->thr = s.gate_thr;
    a.chain.gate_next->mode = s.gate_mode;
  }
  if (done) *a.chain.done = 1;                     // read by every later launch of the chain
  else if (a.chain.sm_src) *a.chain.done = 0;      // (first slot of a warm start: the flag still says the previous solve ended)
  outv[nout] = (double)need;
  outv[nout + 1] = (double)s.phase;
  outv[nout + 2] = (double)(done | (disagree ? 2 : 0));
  for (int k = 0; k < n; k++) {
    const double xk = s.req_at_x ? s.x[k] : s.x1[k];
    outv[nout + 3 + k] = (moved && !done) ? xk : 0.0;
    if (moved && !done) a.chain.x_req[k] = xk;
  }
}

template <int NT, bool CHAIN = false>  // CHAIN: the device-driven solve's variant (the machine's step is compiled in)
__device__ __forceinline__ void finalize_body(const FinalizeArgs &a, FinSmem &sm) {
  constexpr int NW = NT / 64;   // waves
  constexpr int HALF = NW / 2;  // waves per image-moment row
  static_assert(NW >= 2 && NW <= 16 && 2 + 3 * kMaxKnots < NT, "finalize geometry");
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if (t == 0) sm.shchk = 0ull;
  const double N = a.npix;
  const bool chain = CHAIN && a.chain.sm != nullptr;
  constexpr int kSmWords = (int)(sizeof(ChainMachine) / 8);
  static_assert(sizeof(ChainMachine) % 8 == 0 && kSmWords <= NT, "the machine is copied by one thread per word");
  if (chain && t < kSmWords)  // the machine's state (written by the previous evaluation's finalize: an earlier launch) -> LDS
    reinterpret_cast<unsigned long long *>(&sm.csm)[t] =
        reinterpret_cast<const unsigned long long *>(a.chain.sm_src ? a.chain.sm_src : a.chain.sm)[t];
  // Everything this workgroup reads was written by other CUs: each dependent round of loads is a ~1.2 us trip to memory.
  // The reads that do not depend on one another -- fallback counter, accumulator rows, moment rows -- are issued together.
  unsigned fb_count = 0u;
  if (t == 0 && a.fallback) fb_count = *a.fallback;
  const int ncol_acc = (a.gacc && a.measure != 2 && a.gP > 0) ? (a.mu_free ? 2 * a.gP : a.gP) : 0;
  double gv[kTailShards];
  if (t < ncol_acc) {
#pragma unroll
    for (int q = 0; q < kTailShards; q++) gv[q] = ld_sc1(a.gacc + (size_t)q * a.gacc_stride + t);
  }
  if (a.measure != 2 && a.gP > 0 && !a.gacc) {
    // adjoint mode: grad_k = (2/N) (S1_k - mu*S2_k);  gpartials is [column][gblocks], columns = S1 (gP) then S2 (gP).
    // ONE workgroup reads tables other CUs wrote: every load is a ~1-2 us round trip to memory / a remote L2, so what
    // matters is how many are in flight.  Many rows, few columns (front end: ~1000 x 6): all threads stride over the rows
    // with one accumulator per column -- every thread's loads go out in one round.  Few rows, many columns (back end:
    // ~200 x 42): each wave takes whole columns, four at a time.
    const int ncol = a.mu_free ? 2 * a.gP : a.gP;
    if (a.gP > 0 && ncol <= 8) {
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int b0 = 0; b0 < a.gblocks; b0 += 2 * NT) {
        // branch-free: every load of the round is issued before the first use (a predicated load inside the accumulation
        // made the compiler wait for each one: 16 serial ~1 us round trips)
        double v[2][8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b = b0 + u * NT + t;
          const int bb = b < a.gblocks ? b : 0;
#pragma unroll
          for (int j = 0; j < 8; j++) v[u][j] = ld_sc1(a.gpartials + (size_t)(j < ncol ? j : 0) * a.gblocks + bb);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const bool ok = b0 + u * NT + t < a.gblocks;
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] += (ok && j < ncol) ? v[u][j] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double w = wave_sum(acc[j]);
        if (lane == 0) sm.colw[wave][j] = w;
      }
      __syncthreads();
      if (t < ncol) {
        double w = 0;
        for (int q = 0; q < NW; q++) w += sm.colw[q][t];
        sm.cols[t] = w;
      }
    } else {
      for (int k0 = wave; k0 < ncol; k0 += 4 * NW) {  // this wave's columns k0, k0 + NW, k0 + 2 NW, k0 + 3 NW
        double acc[4] = {0, 0, 0, 0};
        const double *r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = a.gpartials + (size_t)min(k0 + j * NW, ncol - 1) * a.gblocks;
        int b = lane;
        for (; b + 192 < a.gblocks; b += 256) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const double v0 = ld_sc1(r[j] + b), v1 = ld_sc1(r[j] + b + 64), v2 = ld_sc1(r[j] + b + 128), v3 = ld_sc1(r[j] + b + 192);
            acc[j] += (v0 + v1) + (v2 + v3);
          }
        }
        for (; b < a.gblocks; b += 64) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] += ld_sc1(r[j] + b);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double v = wave_sum(acc[j]);
          if (lane == 0 && k0 + j * NW < ncol) sm.cols[k0 + j * NW] = v;
        }
      }
    }
    __syncthreads();
  }
  if (CHAIN && a.macc) {  // self-gating slot: the moments are 8 accumulator rows (summed in the order chain_moment_sums uses)
    if (t == 0) chain_moment_sums(a.macc, sm.sh[0], sm.sh[1]);
    if (t < 2 * kTailShards) a.macc_clear[(size_t)(t >> 1) * 16 + (t & 1)] = 0.0;  // the NEXT slot's buffer (nobody reads it now)
  } else if (a.direct) {  // few tiles: sum the image kernel's per-tile moments here instead of a separate launch
    // this is ONE workgroup reading tables other CUs just wrote (L2-remote): keep many independent loads in flight
    const int row = wave & 1, part = wave >> 1;
    double p = 0;
    const double *src = a.partials + (size_t)row * a.nblk;
    const int nvalid = a.nvalid ? (int)(*a.nvalid) : a.nblk;  // list path: only the first *nvalid entries were written
    int b = part * 64 + lane;
    for (; b + 3 * HALF * 64 < nvalid; b += 4 * HALF * 64) {
      const double v0 = ld_sc1(src + b), v1 = ld_sc1(src + b + HALF * 64), v2 = ld_sc1(src + b + 2 * HALF * 64),
                   v3 = ld_sc1(src + b + 3 * HALF * 64);
      p += (v0 + v1) + (v2 + v3);
    }
    for (; b < nvalid; b += HALF * 64) p += ld_sc1(src + b);
    p = wave_sum(p);
    if (lane == 0) sm.shp[wave] = p;
    __syncthreads();
    if (t < 2) {
      double s = 0;
      for (int w = 0; w < HALF; w++) s += sm.shp[2 * w + t];
      sm.sh[t] = s;
    }
  } else if (t < 2) {
    sm.sh[t] = a.sums[t];
  }
  if (t < ncol_acc) {  // accumulator rows (loaded above): their sum, then the zeros the next launch expects
    double w = 0;
#pragma unroll
    for (int q = 0; q < kTailShards; q++) w += gv[q];
    sm.cols[t] = w;
#pragma unroll
    for (int q = 0; q < kTailShards; q++) st_sc1(a.gacc + (size_t)q * a.gacc_stride + t, 0.0);
  }
  __syncthreads();
  const double s0 = sm.sh[0], s1 = sm.sh[1];
  const double mu = s0 / N;
  if (t == 0) {
    double mu_unused;
    const double c = contrast_from_sums(s0, s1, N, a.measure, &mu_unused);
    sm.outv[0] = c;
    sm.outv[1] = mu;
    if (a.gate_out && !chain) *a.gate_out = gate_condition(c, a.gate_thr, a.gate_mode);  // read by the NEXT launch (kernel boundary)
    if (a.fallback) {
      sm.shfall = (double)fb_count;
      *a.fallback = 0u;
    } else {
      sm.shfall = 0.0;
    }
  }
  if (a.measure == 2) {  // gradient magnitude: rows of the Sobel partial table [1+gP][gblocks]
    __syncthreads();     // row 0 overwrites the contrast thread 0 staged above
    for (int k = wave; k < 1 + a.gP; k += NW) {
      double s = 0;
      for (int b = lane; b < a.gblocks; b += 64) s += a.gpartials[(size_t)k * a.gblocks + b];
      s = wave_sum(s);
      if (lane == 0) sm.outv[k == 0 ? 0 : 1 + k] = (k == 0) ? s / N : 2.0 * s / N;
    }
  } else {
    // derivative-plane mode: moments of the P blurred planes were reduced into sums[] by reduce_partials
    for (int k = t; k < a.P; k += NT) {
      const double eD = a.sums[2 + 2 * k] / N, eID = a.sums[3 + 2 * k] / N;
      sm.outv[2 + k] = (a.measure == 1) ? 2.0 * eID : 2.0 * (eID - mu * eD);
    }
    __syncthreads();
    for (int k = t; k < a.gP; k += NT) {
      const double s = sm.cols[k], s2 = a.mu_free ? sm.cols[a.gP + k] : 0.0;
      sm.outv[2 + k] = 2.0 * (s - ((a.mu_free && a.measure != 1) ? mu * s2 : 0.0)) / N;
    }
  }
  __syncthreads();
  int nout = 2 + (a.P > a.gP ? a.P : a.gP);
  if (CHAIN && nout < a.nout_pad) {  // (a fixed layout for the host whatever this launch computed)
    if (t >= nout && t < a.nout_pad) sm.outv[t] = 0.0;
    nout = a.nout_pad;
    __syncthreads();
  }
  if (chain) {
    // Device-driven solve: thread 0 feeds the machine what this evaluation produced -- f = -contrast after a cost evaluation,
    // df = -gradient after a gradient pass -- exactly what the host-driven loop feeds it (cmx_solver.cpp: contrast_fdf), and
    // publishes the next request: the evaluation point for the next slot's kernels, the gate of the gradient pass queued
    // behind this launch, the end of the solve.  The result block carries need / phase / done / next point behind the
    // evaluation's own numbers; the host replays the machine on them and compares (cmx_chain.cpp).
    if (t == 0) {
      // (the machine stays in LDS: a register copy of its ~60 words for the step took the gather to 256 VGPRs + scratch)
      chain_step(a, sm.csm, sm.outv, nout);
    }
    nout += kChainExtra;
    __syncthreads();
    if (t < kSmWords) reinterpret_cast<unsigned long long *>(a.chain.sm)[t] = reinterpret_cast<const unsigned long long *>(&sm.csm)[t];
  }
  // Results, then a checksum and the completion ticket the host spins on (sync_and_collect): the host accepts the
  // results once the ticket matches AND the checksum over what it read matches, so no system-scope fence (an L2
  // write-back, ~3 us here) is needed to order these stores over PCIe -- a torn read simply fails the check and is
  // repeated.  The kernel's end reaches the host through the runtime's completion signal several microseconds later.
  unsigned long long bits = 0ull;
  if (t < nout) {
    const double v = sm.outv[t];
    a.result[t] = v;
    bits = (unsigned long long)__double_as_longlong(v);
  } else if (t == nout) {
    const double v = sm.shfall;
    a.result[kFallbackSlot] = v;
    bits = (unsigned long long)__double_as_longlong(v);
  }
  if (t <= nout) atomicXor(&sm.shchk, bits);
  __syncthreads();
  if (t == 0) {
    // plain stores, no wait between them: the host accepts a snapshot only when ticket AND checksum match what it read, so
    // the order in which these words cross PCIe does not matter (volatile stores made the compiler drain each one)
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(a.result);
    slots[kChecksumSlot] = sm.shchk ^ (a.ticket * kTicketMix);
    slots[kTicketSlot] = a.ticket;
  }
}

template <bool CHAIN>
__global__ __launch_bounds__(1024) void finalize_kernel(FinalizeArgs a) {
  __shared__ FinSmem sm;
  if (CHAIN && a.chain.sm && *a.chain.done) return;  // device-driven solve: finished (the kernels in front returned as well)
  finalize_body<1024, CHAIN>(a, sm);
}

// ---- tail finalize: the evaluation's last kernel runs finalize in its LAST-ARRIVING workgroup instead of handing over to
// a one-workgroup launch (a ~1.7 us kernel boundary + a ~5.4 us kernel whose useful work is a few hundred loads).
// Protocol (cdna_hip_programming.md, Guideline 16): every workgroup stores its partial results WRITE-THROUGH (st_sc1),
// drains them (s_waitcnt vmcnt(0) in every storing wave, then the workgroup barrier) and ONE lane takes a ticket.  One
// counter would serialise ~1000 same-address atomics (~12 ns each); the tickets are therefore sharded by blockIdx % 8
// (= the XCD a workgroup runs on, for speed only -- nothing depends on the placement), and the last arriver of a shard
// takes a ticket on the top counter.  The last arriver overall resets nothing but the counters it completed (they are
// all-zero again when the launch ends), performs ONE agent-scope acquire and runs finalize_body.
// Call from every thread of every workgroup exactly once, after the workgroup's partial results have been issued.
// Returns true in every thread of the one workgroup that must run the finalize.
__device__ __forceinline__ bool tail_arrive(const TailArgs &tl, int nblocks, int block, FinSmem &sm) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nshards = nblocks < kTailShards ? nblocks : kTailShards;
    const int shard = block % kTailShards;
    const unsigned shard_size = (unsigned)((nblocks - shard + kTailShards - 1) / kTailShards);
    unsigned *cs = tl.counters + shard * kTailStride, *ct = tl.counters + kTailShards * kTailStride;
    int last = 0;
    if (atomicAdd(cs, 1u) == shard_size - 1u) {
      __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody else touches it in this launch
      if (atomicAdd(ct, 1u) == (unsigned)nshards - 1u) {
        __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    // no acquire fence (buffer_inv, ~1.7 us): everything the finalize reads from THIS launch was stored write-through and
    // is loaded with sc1 (L1-bypassing, agent-scope) loads -- Guideline 16: "sc1 loads may replace the acquire only when
    // the producer stored sc1"; data of earlier launches is ordered by the kernel boundary
    sm.is_last = last;
  }
  __syncthreads();
  return sm.is_last != 0;
}

// Sharded panoramas: the EXCHANGE SET.  The ranks' votes cover a few per cent of a panorama's tiles, so only those tiles
// of the partial planes travel: flags = the occupancy map all-reduced with max (identical on every rank), cur_member = the set
// the host sized THIS evaluation's exchange for (built by this kernel one evaluation earlier; null: the whole planes travel).
// One workgroup writes
//   next_list / next_member: the flagged tiles dilated by kXsetDx columns (wrapping: the panorama's seam) and kXsetDy rows, in
//                            ascending tile order -- the next evaluation's exchange set (parameters move the votes by pixels);
//   miss_list:               flagged tiles outside cur_member -- what this evaluation's exchange did not cover (a jump of the
//                            parameters); the host completes the evaluation with a second exchange of exactly these;
//   out[0..2] = |next|, |miss|, |flagged|, out[3] = stamp (the words lie outside the finalize's checksummed snapshot).
constexpr int kXsetDx = 1, kXsetDy = 1;
__global__ __launch_bounds__(1024) void xset_kernel(const unsigned char *flags, int tiles_x, int tiles_y, const unsigned char *cur_member,
                                                    int *next_list, unsigned char *next_member, int *miss_list, double *out,
                                                    unsigned long long seq) {
  __shared__ int wave_tot[3][16];
  constexpr int kLdsTiles = 16384;  // 8192 x 2048 pixels; larger maps read the flags through the caches
  __shared__ __attribute__((aligned(16))) unsigned char sflags[kLdsTiles];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = tiles_x * tiles_y;
  // the 3 x 3 neighbourhoods re-read every flag nine times: one coalesced pass brings the map into LDS first (a thread's
  // dependent byte loads from L2 were 18 us of this kernel at 8192 tiles)
  const unsigned char *fl = flags;
  if (n <= kLdsTiles) {
    if ((n & 15) == 0 && (reinterpret_cast<uintptr_t>(flags) & 15) == 0) {
      for (int t = 16 * tid; t < n; t += 16 * 1024) *reinterpret_cast<uint4 *>(sflags + t) = *reinterpret_cast<const uint4 *>(flags + t);
    } else {
      for (int t = tid; t < n; t += 1024) sflags[t] = flags[t];
    }
    __syncthreads();
    fl = sflags;
  }
  // every thread owns a contiguous run of tiles: ONE workgroup-wide scan of the per-thread counts orders the lists
  const int per = (n + 1023) / 1024;
  const int t_beg = min(n, tid * per), t_end = min(n, t_beg + per);
  unsigned nxt_bits = 0, miss_bits = 0;
  int c_nxt = 0, c_miss = 0, c_flag = 0;
  for (int t = t_beg; t < t_end; t++) {
    const int tx = t % tiles_x, ty = t / tiles_x;
    const bool f = fl[t] != 0;
    int nxt = 0;
    for (int dy = -kXsetDy; dy <= kXsetDy; dy++) {
      const int yy = ty + dy;
      if (yy < 0 || yy >= tiles_y) continue;
      for (int dx = -kXsetDx; dx <= kXsetDx; dx++) nxt |= fl[yy * tiles_x + (tx + dx + tiles_x) % tiles_x] != 0;
    }
    const int miss = (f && cur_member && !cur_member[t]) ? 1 : 0;
    next_member[t] = (unsigned char)nxt;
    if (t - t_beg < 32) { nxt_bits |= (unsigned)nxt << (t - t_beg); miss_bits |= (unsigned)miss << (t - t_beg); }
    c_nxt += nxt; c_miss += miss; c_flag += f ? 1 : 0;
  }
  int inc0 = c_nxt, inc1 = c_miss, inc2 = c_flag;  // inclusive scans inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v0 = __shfl_up(inc0, o, 64), v1 = __shfl_up(inc1, o, 64), v2 = __shfl_up(inc2, o, 64);
    if (lane >= o) { inc0 += v0; inc1 += v1; inc2 += v2; }
  }
  if (lane == 63) { wave_tot[0][wave] = inc0; wave_tot[1][wave] = inc1; wave_tot[2][wave] = inc2; }
  __syncthreads();
  int off0 = inc0 - c_nxt, off1 = inc1 - c_miss, tot0 = 0, tot1 = 0, tot2 = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wave) { off0 += wave_tot[0][w]; off1 += wave_tot[1][w]; }
    tot0 += wave_tot[0][w];
    tot1 += wave_tot[1][w];
    tot2 += wave_tot[2][w];
  }
  for (int t = t_beg; t < t_end; t++) {
    int nxt, miss;
    if (t - t_beg < 32) {
      nxt = (nxt_bits >> (t - t_beg)) & 1u;
      miss = (miss_bits >> (t - t_beg)) & 1u;
    } else {  // (maps beyond 32768 tiles: read back what the first pass stored / recompute)
      nxt = next_member[t];
      miss = (fl[t] != 0 && cur_member && !cur_member[t]) ? 1 : 0;
    }
    if (nxt) next_list[off0++] = t;
    if (miss) miss_list[off1++] = t;
  }
  if (tid == 0) {
    const double w0 = (double)tot0, w1 = (double)tot1, w2 = (double)tot2;
    out[0] = w0;
    out[1] = w1;
    out[2] = w2;
    reinterpret_cast<unsigned long long *>(out)[3] = (unsigned long long)__double_as_longlong(w0) ^ (unsigned long long)__double_as_longlong(w1) ^
                                                      (unsigned long long)__double_as_longlong(w2) ^ (seq * kTicketMix);
  }
}
void launch_xset(const unsigned char *flags, int tiles_x, int tiles_y, const unsigned char *cur_member, int *next_list,
                 unsigned char *next_member, int *miss_list, double *out, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(xset_kernel, dim3(1), dim3(1024), 0, s, flags, tiles_x, tiles_y, cur_member, next_list, next_member, miss_list, out,
                     seq);
}

// the listed tiles of both planes <-> one contiguous staging buffer [plane][entry][kTileY][kTileX] (pixels beyond the image: zero);
// with `flags`, the occupancy map rides behind them as floats (summed: > 0 = some rank voted there), so that the map and the
// tiles are ONE collective -- blockIdx.y == 2 are the map's workgroups
template <bool UNPACK>
__global__ __launch_bounds__(256) void xset_copy_kernel(float *planes, size_t np, int W, int H, int tiles_x, const int *list, int n,
                                                        float *stage, unsigned char *flags, int ntiles) {
  if (blockIdx.y == 2) {
    float *sf = stage + (size_t)2 * n * (kTileX * kTileY);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
      if (UNPACK) flags[t] = sf[t] > 0.f ? 1 : 0;
      else sf[t] = flags[t] ? 1.f : 0.f;
    }
    return;
  }
  if ((int)blockIdx.x >= n) return;
  const int i = blockIdx.x, plane = blockIdx.y;
  const int tile = list[i];
  const int x0 = (tile % tiles_x) * kTileX, y0 = (tile / tiles_x) * kTileY;
  float *img = planes + (size_t)plane * np;
  float *st = stage + ((size_t)plane * n + i) * (kTileX * kTileY);
  for (int p = threadIdx.x; p < kTileX * kTileY; p += 256) {
    const int lx = p % kTileX, ly = p / kTileX;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool in = gx < W && gy < H;
    if (UNPACK) {
      if (in) img[(size_t)gy * W + gx] = st[p];
    } else {
      st[p] = in ? img[(size_t)gy * W + gx] : 0.f;
    }
  }
}
void launch_xset_copy(bool unpack, float *planes, size_t np, int W, int H, const int *list, int n, float *stage, unsigned char *flags,
                      int ntiles, hipStream_t s) {
  if (n <= 0 && !flags) return;
  const int tiles_x = (W + kTileX - 1) / kTileX;
  const dim3 grid(n > 0 ? n : 1, flags ? 3 : 2);
  if (unpack) hipLaunchKernelGGL(xset_copy_kernel<true>, grid, dim3(256), 0, s, planes, np, W, H, tiles_x, list, n, stage, flags, ntiles);
  else hipLaunchKernelGGL(xset_copy_kernel<false>, grid, dim3(256), 0, s, planes, np, W, H, tiles_x, list, n, stage, flags, ntiles);
}

// The group's one-shot exchange fused with the unpack: every member reads ALL members' packed send buffers (peer pointers; 16-byte loads)
// and writes the sum -- added in member order: the same bits on every member -- straight into its own planes and occupancy map.  No
// receive buffer, no separate unpack launch (cmx_group.cpp: direct_peers; cmx_comm.cpp: exchange_tiles).
__global__ __launch_bounds__(256) void xset_sum_unpack_kernel(XsetPeers in, float *planes, size_t np, int W, int H, int tiles_x, const int *list,
                                                              int n, unsigned char *flags, int ntiles) {
  if (in.xdev) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // peers on other devices: this device's L2 may hold their buffers' old lines
  if (blockIdx.y == 2) {
    const size_t off = (size_t)2 * n * (kTileX * kTileY);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
      float v = in.p[0][off + t];
      for (int m = 1; m < in.n; m++) v += in.p[m][off + t];
      flags[t] = v > 0.f ? 1 : 0;
    }
    return;
  }
  if ((int)blockIdx.x >= n) return;
  const int i = blockIdx.x, plane = blockIdx.y;
  const int tile = list[i];
  const int x0 = (tile % tiles_x) * kTileX, y0 = (tile / tiles_x) * kTileY;
  float *img = planes + (size_t)plane * np;
  const size_t base = ((size_t)plane * n + i) * (kTileX * kTileY);
  static_assert(kTileX % 4 == 0 && (kTileX * kTileY) % 1024 == 0, "one float4 per thread per pass");
  for (int p = threadIdx.x * 4; p < kTileX * kTileY; p += 1024) {
    float4 acc = *reinterpret_cast<const float4 *>(in.p[0] + base + p);
    for (int m = 1; m < in.n; m++) {
      const float4 v = *reinterpret_cast<const float4 *>(in.p[m] + base + p);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int lx = p % kTileX, ly = p / kTileX;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gy < H) {
      float *dst = img + (size_t)gy * W + gx;
      if (gx + 3 < W && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) *reinterpret_cast<float4 *>(dst) = acc;
      else {
        if (gx < W) dst[0] = acc.x;
        if (gx + 1 < W) dst[1] = acc.y;
        if (gx + 2 < W) dst[2] = acc.z;
        if (gx + 3 < W) dst[3] = acc.w;
      }
    }
  }
}
void launch_xset_sum_unpack(const XsetPeers &in, float *planes, size_t np, int W, int H, const int *list, int n, unsigned char *flags,
                            int ntiles, hipStream_t s) {
  if (n <= 0 && !flags) return;
  const int tiles_x = (W + kTileX - 1) / kTileX;
  const dim3 grid(n > 0 ? n : 1, flags ? 3 : 2);
  hipLaunchKernelGGL(xset_sum_unpack_kernel, grid, dim3(256), 0, s, in, planes, np, W, H, tiles_x, list, n, flags, ntiles);
}

size_t image_lds_bytes(int r) {
  const int rawW = kTileX + 2 * r, rawH = kTileY + 2 * r;
  return sizeof(float) * ((size_t)rawW * rawH + (size_t)kTileX * rawH) + sizeof(double) * 8;
}

// One workgroup = one 64x16 output tile x one group of <= kPlaneGroup derivative planes (blockIdx.z).
// LDS: raw tile with halo -> row-blurred tile -> column pass in registers -> fp64 moments.
// Row pass:  s = k[0]*S[x-r]; s += k[j]*S[x-r+j]           (generic row filter order)
// Col pass:  s = k[r]*T[y];   s += k[r+j]*(T[y+j]+T[y-j])  (symmetric column filter order)
// LIST: walk the compacted tile work list with a bounded grid (large panoramas); otherwise one tile per workgroup and
// the loop below runs exactly once (kept as a loop so that both forms share one body; a runtime trip count costs the
// one-tile form its register allocation -- the taps spill -- hence the compile-time switch)
template <int R, bool LIST>  // R >= 0: compile-time radius (loops unroll, taps live in registers, index maths is constant); -1: a.r
__device__ __forceinline__ void image_moments_body(const ImgArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ FinSmem fin_sm;  // tail finalize scratch (sizeof % 16 == 0: the dynamic region stays aligned)
  const bool tail = a.tail.counters != nullptr;  // P == 0 by construction (host side)
  const int r = (R >= 0) ? R : a.r;
  const int W = a.W, H = a.H;
  float taps[2 * kMaxRadius + 1];
#pragma unroll
  for (int j = 0; j < 2 * kMaxRadius + 1; j++) taps[j] = (R < 0 || j <= 2 * R) ? a.taps[j] : 0.f;
  const int rawW = kTileX + 2 * r, rawH = kTileY + 2 * r;
  double *red = reinterpret_cast<double *>(smem_raw);
  float *raw = reinterpret_cast<float *>(smem_raw + 8 * sizeof(double));
  float *rowb = raw + rawW * rawH;
  const int tid = threadIdx.x;
  const int tx = tid & 63, tq = tid >> 6;  // output column, row quad
  const size_t np = (size_t)W * H;
  const float alpha = a.alpha ? (float)(*a.alpha) : 0.f;
  const int g = blockIdx.z;                // plane group; group 0 also owns the I moments
  const int k_beg = g * kPlaneGroup;
  const int k_end = min(a.P, k_beg + kPlaneGroup);
  const int n_work = LIST ? (int)(*a.tile_count) : 0;
  for (int wi = blockIdx.x, once = 1; LIST ? (wi < n_work) : (once != 0); wi += gridDim.x, once = 0) {
  const unsigned entry = LIST ? a.tile_list[wi] : (unsigned)wi;
  const int tile = (int)(entry & 0x3fffffffu);
  const int x0 = (tile % a.tiles_x) * kTileX, y0 = (tile / a.tiles_x) * kTileY;
  if (LIST) __syncthreads();  // LDS of the previous tile is free

  float I[4] = {0.f, 0.f, 0.f, 0.f};
  bool valid[4];
#pragma unroll
  for (int j = 0; j < 4; j++) valid[j] = (x0 + tx < W) && (y0 + tq * 4 + j < H);
  if (a.zero_ptr && g == 0) {  // clear this tile of the other accumulation buffer (nobody reads it during this launch)
    const bool dirty = LIST ? (entry & 0x40000000u) != 0 : (!a.flags_other || a.flags_other[tile] != 0);
    if (dirty) {
      for (int pl = 0; pl < a.zero_planes; pl++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (valid[j]) a.zero_ptr[(size_t)pl * np + (size_t)(y0 + tq * 4 + j) * W + (x0 + tx)] = 0.f;
    }
    if (!LIST) {  // (the list pre-pass has already un-flagged it)
      __syncthreads();   // every thread has read the flag
      if (tid == 0 && a.flags_other && dirty) a.flags_other[tile] = 0;
    }
  }
  const int slot = LIST ? wi : tile;  // row position of this tile's partial moments
  if (LIST ? !(entry & 0x80000000u) : !tile_active(a, tile % a.tiles_x, tile / a.tiles_x, r, kTileX, kTileY)) {
    if (tid == 0 && g == 0) {  // nothing within reach: all sums are zero
      if (tail) {
        st_sc1(a.partials + (size_t)0 * a.nblk + slot, 0.0);
        st_sc1(a.partials + (size_t)1 * a.nblk + slot, 0.0);
      } else {
        a.partials[(size_t)0 * a.nblk + slot] = 0.0;
        a.partials[(size_t)1 * a.nblk + slot] = 0.0;
      }
    }
    continue;
  }

  // pass over plane 0 (k == -1) then the group's derivative planes
  for (int k = -1; k < k_end; k = (k < 0 ? k_beg : k + 1)) {
    __syncthreads();
    for (int idx = tid; idx < rawW * rawH; idx += kImgThreads) {
      const int ly = idx / rawW, lx = idx - ly * rawW;
      const int gx = reflect101(x0 + lx - r, W), gy = reflect101(y0 + ly - r, H);
      const size_t off = (size_t)gy * W + gx;
      float v;
      if (k < 0) {
        v = a.src_a[off];
        if (a.src_b) v = v + a.src_b[off];        // IL = IL_old + IL_new
        if (a.igp) v = a.igp[off] * alpha + v;     // I = IGp*alpha + IL
      } else {
        v = a.dplanes[(size_t)k * np + off];
      }
      raw[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < kTileX * rawH; idx += kImgThreads) {
      const int ly = idx >> 6, lx = idx & 63;
      const float *S = raw + ly * rawW + lx;
      float s = taps[0] * S[0];
#pragma unroll
      for (int j = 1; j <= 2 * r; j++) s += taps[j] * S[j];
      rowb[idx] = s;
    }
    __syncthreads();
    double sD = 0, sID = 0, sI = 0, sII = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int ly = tq * 4 + j + r;
      const float *T = rowb + ly * kTileX + tx;
      float s = taps[r] * T[0];
#pragma unroll
      for (int t = 1; t <= r; t++) s += taps[r + t] * (T[t * kTileX] + T[-t * kTileX]);
      if (valid[j]) {
        const size_t o = (size_t)(y0 + tq * 4 + j) * W + (x0 + tx);
        if (k < 0) {
          I[j] = s;
          sI += (double)s;
          sII += (double)s * (double)s;
          if (a.out_blur0 && g == 0) a.out_blur0[o] = s;
        } else {
          sD += (double)s;
          sID += (double)I[j] * (double)s;
          if (a.out_blurd) a.out_blurd[(size_t)k * np + o] = s;
        }
      }
    }
    if (k < 0) {
      if (g == 0) {
        double t0, t1;
        block_sum2(sI, sII, red, kImgThreads / 64, t0, t1);
        if (tid == 0) {
          if (tail) {  // write-through: the last-arriving workgroup of this launch reads them
            st_sc1(a.partials + (size_t)0 * a.nblk + slot, t0);
            st_sc1(a.partials + (size_t)1 * a.nblk + slot, t1);
          } else {
            a.partials[(size_t)0 * a.nblk + slot] = t0;
            a.partials[(size_t)1 * a.nblk + slot] = t1;
          }
        }
      }
    } else {
      double t0, t1;
      block_sum2(sD, sID, red, kImgThreads / 64, t0, t1);
      if (tid == 0) {
        a.partials[(size_t)(2 + 2 * k) * a.nblk + tile] = t0;
        a.partials[(size_t)(3 + 2 * k) * a.nblk + tile] = t1;
      }
    }
  }
  }  // work loop
  if (tail && tail_arrive(a.tail, (int)gridDim.x, (int)blockIdx.x, fin_sm)) finalize_body<kImgThreads>(a.tail.fin, fin_sm);
}
template <int R, bool LIST>
__global__ __launch_bounds__(kImgThreads) void image_moments_kernel(ImgArgs a) { image_moments_body<R, LIST>(a); }
// (an occupancy bound of 8 waves per SIMD -- 54 instead of 82 VGPRs, two workgroups per CU -- made this kernel SLOWER:
// 9.7 -> 10.5 us with the tail finalize at 640x480; image_adjoint2_kernel is the opposite case)

void launch_image_moments(const ImgArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const int groups = a.P > 0 ? (a.P + kPlaneGroup - 1) / kPlaneGroup : 1;
  const size_t lds = image_lds_bytes(a.r);
  if (a.tile_list) {
    const dim3 g(min(a.nblk, kTileListGrid), 1, groups);
    if (a.r == 4) CMX_LAUNCH((image_moments_kernel<4, true>), g, dim3(kImgThreads), lds, s, t0, t1, a);
    else CMX_LAUNCH((image_moments_kernel<-1, true>), g, dim3(kImgThreads), lds, s, t0, t1, a);
  } else {
    const dim3 g(a.nblk, 1, groups);
    if (a.r == 4) CMX_LAUNCH((image_moments_kernel<4, false>), g, dim3(kImgThreads), lds, s, t0, t1, a);
    else CMX_LAUNCH((image_moments_kernel<-1, false>), g, dim3(kImgThreads), lds, s, t0, t1, a);
  }
}

// ---------------------------------------------------------------------------------------------- reduce + finalize
__global__ __launch_bounds__(256) void reduce_partials_kernel(FinalizeArgs a) {
  __shared__ double red[4];
  const int q = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.partials[(size_t)q * a.nblk + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) a.sums[q] = s;
}

// per-parameter sum of the gather kernel's block partials -> gsum[P] (the buffer ranks all-reduce)
__global__ __launch_bounds__(256) void reduce_gpartials_kernel(const double *gpartials, int gblocks, int P, double *gsum) {
  __shared__ double red[4];
  const int k = blockIdx.x;
  double s = 0;
  for (int b = threadIdx.x; b < gblocks; b += 256) s += gpartials[(size_t)k * gblocks + b];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gsum[k] = s;
}
void launch_reduce_gpartials(const double *gpartials, int gblocks, int P, double *gsum, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(reduce_gpartials_kernel, dim3(P), dim3(256), 0, s, gpartials, gblocks, P, gsum);
}

void launch_reduce_partials(const FinalizeArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 + 2 * a.P), dim3(256), 0, s, a);
}
void launch_finalize_only(const FinalizeArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (a.chain.sm) CMX_LAUNCH(finalize_kernel<true>, dim3(1), dim3(1024), 0, s, t0, t1, a);
  else CMX_LAUNCH(finalize_kernel<false>, dim3(1), dim3(1024), 0, s, t0, t1, a);
}
void launch_finalize(const FinalizeArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  launch_reduce_partials(a, s);
  launch_finalize_only(a, s, t0, t1);
}

// ---------------------------------------------------------------------------------------------- fused image + adjoint
// One workgroup = one TX x TY tile of Jt = G^T B^.  It needs B on the tile + r halo, hence the raw image on the tile +
// 2r halo; the moments of B are taken over the tile's own pixels only, so every pixel is counted once.
// Tile shape / workgroup size are template parameters (swept on MI355X with tools/sweep_image_tile.sh: < 10 % effect).
constexpr int kAdjTX = 64, kAdjTY = 16, kAdjThreads = 1024;
static_assert(kAdjTX == kTileX && kAdjTY == kTileY, "the tile-occupancy flags are shared by both image kernels");

size_t image_adjoint_lds_bytes(int r) {
  const size_t aw = kAdjTX + 4 * r, ah = kAdjTY + 4 * r, bw = kAdjTX + 2 * r, bh = kAdjTY + 2 * r;
  return sizeof(double) * 32 + sizeof(float) * (aw * ah + bw * ah + bw * bh + (size_t)kAdjTX * bh);
}
int image_adjoint_tiles_x(int W) { return (W + kAdjTX - 1) / kAdjTX; }
int image_adjoint_tiles(int W, int H) { return image_adjoint_tiles_x(W) * ((H + kAdjTY - 1) / kAdjTY); }

// block-wide sum for NT threads; result valid in every thread.  red: 16 doubles of LDS.
__device__ __forceinline__ double block_sum_n(double v, double *red, int nwaves) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < nwaves; w++) s += red[w];
  return s;
}

template <int R, int TX, int TY, int NT, bool LIST>  // LIST: see image_moments_kernel
__global__ __launch_bounds__(NT) void image_adjoint_kernel(ImgAdjArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (wg_stop_requested(g.img.skip)) return;  // device-driven solve: finished
  const ImgArgs &a = g.img;
  const int r = (R >= 0) ? R : a.r;
  const int W = a.W, H = a.H;
  float taps[2 * kMaxRadius + 1];
#pragma unroll
  for (int j = 0; j < 2 * kMaxRadius + 1; j++) taps[j] = (R < 0 || j <= 2 * R) ? a.taps[j] : 0.f;
  const int aw = TX + 4 * r, ah = TY + 4 * r, bw = TX + 2 * r, bh = TY + 2 * r;
  double *red = reinterpret_cast<double *>(smem_raw);
  float *bufA = reinterpret_cast<float *>(smem_raw + 32 * sizeof(double));  // raw, aw x ah
  float *bufR = bufA + aw * ah;                                             // row-blurred raw, bw x ah
  float *bufB = bufR + bw * ah;                                             // B^ (0 outside the image), bw x bh
  float *bufT = bufB + bw * bh;                                             // row pass of G^T, TX x bh
  const int tid = threadIdx.x;
  const float alpha = a.alpha ? (float)(*a.alpha) : 0.f;
  const int n_work = LIST ? (int)(*a.tile_count) : 0;
  for (int wi = blockIdx.x, once = 1; LIST ? (wi < n_work) : (once != 0); wi += gridDim.x, once = 0) {
  const unsigned entry = LIST ? a.tile_list[wi] : (unsigned)wi;
  const int tile = (int)(entry & 0x3fffffffu);
  const int x0 = (tile % a.tiles_x) * TX, y0 = (tile / a.tiles_x) * TY;
  if (LIST) __syncthreads();  // LDS of the previous tile is free
  if (a.zero_ptr) {  // clear this tile of the other accumulation buffer (ping-pong: no memset launch next time)
    const bool dirty = LIST ? (entry & 0x40000000u) != 0 : (!a.flags_other || a.flags_other[tile] != 0);
    if (dirty) {
      for (int idx = tid; idx < TX * TY * a.zero_planes; idx += NT) {
        const int pl = idx / (TX * TY), q = idx - pl * (TX * TY);
        const int gx = x0 + (q % TX), gy = y0 + (q / TX);
        if (gx < W && gy < H) a.zero_ptr[(size_t)pl * W * H + (size_t)gy * W + gx] = 0.f;
      }
    }
    if (!LIST) {  // (the list pre-pass has already un-flagged it)
      __syncthreads();   // every thread has read the flag
      if (tid == 0 && a.flags_other && dirty) a.flags_other[tile] = 0;
    }
  }
  // nothing non-zero within 2r of this tile: B and Jt vanish on it, and no vote cell (the only place the gather
  // reads Jt) lies in it -- leave Jt untouched, contribute zero moments
  const int slot = LIST ? wi : tile;  // row position of this tile's partial moments
  if (LIST ? !(entry & 0x80000000u) : !tile_active(a, tile % a.tiles_x, tile / a.tiles_x, 2 * r, TX, TY)) {
    if (tid == 0) {
      a.partials[(size_t)0 * a.nblk + slot] = 0.0;
      a.partials[(size_t)1 * a.nblk + slot] = 0.0;
    }
    continue;
  }

  for (int idx = tid; idx < aw * ah; idx += NT) {
    const int ly = idx / aw, lx = idx - ly * aw;
    const int gx = reflect101(x0 + lx - 2 * r, W), gy = reflect101(y0 + ly - 2 * r, H);
    const size_t off = (size_t)gy * W + gx;
    float v = a.src_a[off];
    if (a.src_b) v = v + a.src_b[off];
    if (a.igp) v = a.igp[off] * alpha + v;
    bufA[idx] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < bw * ah; idx += NT) {  // forward row pass (same op order as image_moments)
    const int ly = idx / bw, lx = idx - ly * bw;
    const float *S = bufA + ly * aw + lx;
    float s = taps[0] * S[0];
#pragma unroll
    for (int j = 1; j <= 2 * r; j++) s += taps[j] * S[j];
    bufR[idx] = s;
  }
  __syncthreads();
  double sI = 0, sII = 0;
  for (int idx = tid; idx < bw * bh; idx += NT) {  // forward column pass -> B on tile + r halo
    const int ly = idx / bw, lx = idx - ly * bw;
    const float *T = bufR + (ly + r) * bw + lx;
    float s = taps[r] * T[0];
#pragma unroll
    for (int t = 1; t <= r; t++) s += taps[r + t] * (T[t * bw] + T[-t * bw]);
    const int gx = x0 + lx - r, gy = y0 + ly - r;
    const bool inside = gx >= 0 && gx < W && gy >= 0 && gy < H;
    bufB[idx] = inside ? s : 0.f;
    if (inside && lx >= r && lx < r + TX && ly >= r && ly < r + TY) {
      sI += (double)s;
      sII += (double)s * (double)s;
      if (a.out_blur0) a.out_blur0[(size_t)gy * W + gx] = s;
    }
  }
  {
    double t0, t1;
    block_sum2(sI, sII, red, NT / 64, t0, t1);
    if (tid == 0) {
      a.partials[(size_t)0 * a.nblk + slot] = t0;
      a.partials[(size_t)1 * a.nblk + slot] = t1;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < TX * bh; idx += NT) {  // G^T row pass: zero-padded conv + folded reflections
    const int ly = idx / TX, lx = idx - ly * TX;
    const float *S = bufB + ly * bw + lx;
    float s = taps[0] * S[0];
#pragma unroll
    for (int j = 1; j <= 2 * r; j++) s += taps[j] * S[j];
    const int gx = x0 + lx;
    const float *Srow = bufB + ly * bw;  // column of global x is (x - x0 + r)
    if (1 <= gx && gx <= r)
      for (int m = 0; m <= r - gx; m++) s += taps[r + gx + m] * Srow[m - x0 + r];
    if (W - 1 - r <= gx && gx <= W - 2) {
      const int d = W - 1 - gx;
      for (int m = 0; m <= r - d; m++) s += taps[r + d + m] * Srow[(W - 1 - m) - x0 + r];
    }
    bufT[idx] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < TX * TY; idx += NT) {
    const int ty = idx / TX, tx = idx - ty * TX;
    const int gy = y0 + ty, gx = x0 + tx;
    if (gx < W && gy < H) {
      const int ly = ty + r;
      const float *T = bufT + ly * TX + tx;
      float s = taps[r] * T[0];
#pragma unroll
      for (int t = 1; t <= r; t++) s += taps[r + t] * (T[t * TX] + T[-t * TX]);
      const float *Tcol = bufT + tx;  // row of global y is (y - y0 + r)
      if (1 <= gy && gy <= r)
        for (int m = 0; m <= r - gy; m++) s += taps[r + gy + m] * Tcol[(m - y0 + r) * TX];
      if (H - 1 - r <= gy && gy <= H - 2) {
        const int d = H - 1 - gy;
        for (int m = 0; m <= r - d; m++) s += taps[r + d + m] * Tcol[((H - 1 - m) - y0 + r) * TX];
      }
      g.jt[(size_t)gy * W + gx] = s;
    }
  }
  }  // work loop
}

// The same pass with G^T G applied as ONE banded operator per axis (Mx, My: rows of M = G^T G with the reflections and the
// restriction to the image folded in, cmx_context.cpp upload_gt1): Jt = My (Mx I) needs no B on a halo, so the tile goes
//   raw (tile + 2r) -> [row pass: G_x raw on tile rows +-r, M_x raw on tile rows +-2r] -> [column pass: B and Jt on the tile]
// -- three barrier-separated phases instead of five.  B (and with it the moments) is computed in the operation order of
// image_moments_kernel; Jt differs from the four-pass form by fp32 rounding only (~1e-7 relative).
size_t image_adjoint2_lds_bytes(int r) {
  const size_t aw = kAdjTX + 4 * r, ah = kAdjTY + 4 * r;
  return sizeof(double) * 32 + sizeof(float) * (aw * ah + (size_t)kAdjTX * (kAdjTY + 2 * r) + (size_t)kAdjTX * ah);
}

// TAIL: finalize in the last-arriving workgroup (cost-only evaluations that keep Jt for the df that follows).  Two 1024-thread
// workgroups fit a CU only at <= 64 VGPRs (8 waves per SIMD), and 300 tiles on 256 CUs need the second one: at 76 VGPRs the
// pass took 11 us instead of 8.2 -- hence the occupancy bound (the finalize body, run by one workgroup, may spill; the
// image phases do not) and the compile-time switch.
template <int R, bool LIST, int TAIL>  // TAIL: 0 none, 1 finalize in the last-arriving workgroup, 2 the same with the device-driven solve's
                                       // step, 3 no finalize: the tile's moments go to accumulator rows (self-gating slots)
__global__ __launch_bounds__(kAdjThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void image_adjoint2_kernel(ImgAdjArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ FinSmem fin_sm;
  if (TAIL >= 2 && wg_stop_requested(g.img.skip)) return;  // device-driven solve: finished
  constexpr bool tail = TAIL == 1 || TAIL == 2;
  constexpr int TX = kAdjTX, TY = kAdjTY, NT = kAdjThreads, NTAP = 4 * R + 1;
  constexpr int AW = TX + 4 * R, AH = TY + 4 * R, GH = TY + 2 * R;
  static_assert(NT == TX * TY, "one thread per tile pixel in the column pass");
  static_assert(AH <= 2 * TY, "two row-pass rows per thread");
  const ImgArgs &a = g.img;
  const int W = a.W, H = a.H;
  float taps[2 * R + 1];
#pragma unroll
  for (int j = 0; j < 2 * R + 1; j++) taps[j] = a.taps[j];
  double *red = reinterpret_cast<double *>(smem_raw);
  float *bufA = reinterpret_cast<float *>(smem_raw + 32 * sizeof(double));  // raw, AW x AH
  float *bufG = bufA + AW * AH;                                             // G_x raw, TX x GH (tile rows -r .. TY+r)
  float *bufM = bufG + TX * GH;                                             // M_x raw, TX x AH (tile rows -2r .. TY+2r)
  const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid / TX;
  const float alpha = a.alpha ? (float)(*a.alpha) : 0.f;
  const int n_work = LIST ? (int)(*a.tile_count) : 0;
  for (int wi = blockIdx.x, once = 1; LIST ? (wi < n_work) : (once != 0); wi += gridDim.x, once = 0) {
  const unsigned entry = LIST ? a.tile_list[wi] : (unsigned)wi;
  const int tile = (int)(entry & 0x3fffffffu);
  const int x0 = (tile % a.tiles_x) * TX, y0 = (tile / a.tiles_x) * TY;
  if (LIST) __syncthreads();  // LDS of the previous tile is free
  if (a.zero_ptr) {  // clear this tile of the other accumulation buffer (ping-pong: no memset launch next time)
    const bool dirty = LIST ? (entry & 0x40000000u) != 0 : (!a.flags_other || a.flags_other[tile] != 0);
    if (dirty) {
      for (int idx = tid; idx < TX * TY * a.zero_planes; idx += NT) {
        const int pl = idx / (TX * TY), q = idx - pl * (TX * TY);
        const int gx = x0 + (q % TX), gy = y0 + (q / TX);
        if (gx < W && gy < H) a.zero_ptr[(size_t)pl * W * H + (size_t)gy * W + gx] = 0.f;
      }
    }
    if (!LIST) {  // (the list pre-pass has already un-flagged it)
      __syncthreads();   // every thread has read the flag
      if (tid == 0 && a.flags_other && dirty) a.flags_other[tile] = 0;
    }
  }
  const int slot = LIST ? wi : tile;  // row position of this tile's partial moments
  if (LIST ? !(entry & 0x80000000u) : !tile_active(a, tile % a.tiles_x, tile / a.tiles_x, 2 * R, TX, TY)) {
    if (tid == 0) {
      if (tail) {
        st_sc1(a.partials + (size_t)0 * a.nblk + slot, 0.0);
        st_sc1(a.partials + (size_t)1 * a.nblk + slot, 0.0);
      } else {
        a.partials[(size_t)0 * a.nblk + slot] = 0.0;
        a.partials[(size_t)1 * a.nblk + slot] = 0.0;
      }
    }
    continue;
  }
  // coefficient rows: every row of M at least 2r away from the border is the same 4r+1-tap kernel (wave-uniform loads);
  // tiles that touch the left / right border fetch the row of their own column, top / bottom the row of the wave's line
  float mx[NTAP];
  {
    const bool interior = x0 >= 2 * R && x0 + TX - 1 <= W - 1 - 2 * R;
    const float *row = g.Mx + (size_t)(interior ? 2 * R : min(x0 + tx, W - 1)) * NTAP;
#pragma unroll
    for (int i = 0; i < NTAP; i++) mx[i] = row[i];
  }
  for (int idx = tid; idx < AW * AH; idx += NT) {
    const int ly = idx / AW, lx = idx - ly * AW;
    const int gx = reflect101(x0 + lx - 2 * R, W), gy = reflect101(y0 + ly - 2 * R, H);
    const size_t off = (size_t)gy * W + gx;
    float v = a.src_a[off];
    if (a.src_b) v = v + a.src_b[off];
    if (a.igp) v = a.igp[off] * alpha + v;
    bufA[idx] = v;  // (beyond the image: the reflected value for G_x; M's rows carry zeros there)
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; h++) {  // row pass: raw rows ty and ty + TY, output column tx
    const int ly = ty + h * TY;
    if (ly < AH) {
      const float *S = bufA + ly * AW + tx;
      float in[NTAP];
#pragma unroll
      for (int i = 0; i < NTAP; i++) in[i] = S[i];
      // 17-term sums of the composite operator in fp64, rounded once: the four-pass form rounds after every 9-term pass,
      // and near a stationary point the gradient is a small difference of these (exact-arithmetic sweep)
      double m = (double)mx[0] * (double)in[0];
#pragma unroll
      for (int i = 1; i < NTAP; i++) m = __builtin_fma((double)mx[i], (double)in[i], m);  // (not an oracle-ordered sum: fused)
      bufM[ly * TX + tx] = (float)m;
      if (ly >= R && ly < R + GH) {  // forward row pass, same op order as image_moments
        float s = taps[0] * in[R];
#pragma unroll
        for (int j = 1; j <= 2 * R; j++) s += taps[j] * in[R + j];
        bufG[(ly - R) * TX + tx] = s;
      }
    }
  }
  float my[NTAP];
  {
    const bool interior = y0 >= 2 * R && y0 + TY - 1 <= H - 1 - 2 * R;
    const float *row = g.My + (size_t)(interior ? 2 * R : min(y0 + ty, H - 1)) * NTAP;  // wave-uniform (a wave is one tile row)
#pragma unroll
    for (int i = 0; i < NTAP; i++) my[i] = row[i];
  }
  __syncthreads();
  double sI = 0, sII = 0;
  {
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
      const float *T = bufG + (ty + R) * TX + tx;
      float s = taps[R] * T[0];
#pragma unroll
      for (int t = 1; t <= R; t++) s += taps[R + t] * (T[t * TX] + T[-t * TX]);
      sI = (double)s;
      sII = (double)s * (double)s;
      if (a.out_blur0) a.out_blur0[(size_t)gy * W + gx] = s;
      const float *Q = bufM + ty * TX + tx;
      double j = (double)my[0] * (double)Q[0];
#pragma unroll
      for (int i = 1; i < NTAP; i++) j = __builtin_fma((double)my[i], (double)Q[i * TX], j);
      g.jt[(size_t)gy * W + gx] = (float)j;
    }
  }
  {
    double t0, t1;
    block_sum2(sI, sII, red, NT / 64, t0, t1);
    if (tid == 0) {
      if (TAIL == 3) {  // accumulator rows: read by every workgroup of the gradient pass queued behind this launch
        double *row = a.macc + (size_t)(slot % kTailShards) * 16;
        if (t0 != 0.0) __hip_atomic_fetch_add(row, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t1 != 0.0) __hip_atomic_fetch_add(row + 1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (tail) {  // write-through: the last-arriving workgroup of this launch reads them
        st_sc1(a.partials + (size_t)0 * a.nblk + slot, t0);
        st_sc1(a.partials + (size_t)1 * a.nblk + slot, t1);
      } else {
        a.partials[(size_t)0 * a.nblk + slot] = t0;
        a.partials[(size_t)1 * a.nblk + slot] = t1;
      }
    }
  }
  }  // work loop
  if (tail && tail_arrive(a.tail, (int)gridDim.x, (int)blockIdx.x, fin_sm)) finalize_body<NT, TAIL == 2>(a.tail.fin, fin_sm);
}

// The same three phases for any blur radius (1 <= r <= kMaxRadius; sigma = 2, 3 -> r = 8, 12): loops over the taps at run
// time, coefficient rows read through the caches instead of living in registers.  Written for accuracy, not speed: with
// 17- and 25-tap kernels the four-pass form rounds four long fp32 sums per pixel, and near a stationary point the gradient is a
// small difference of them -- here Jt is two fp64-accumulated sums (the production path stays within
// 1e-5 of the exact-arithmetic value where the fp32 reference arithmetic itself does not).
template <bool LIST>
__global__ __launch_bounds__(kAdjThreads) void image_adjoint2g_kernel(ImgAdjArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (wg_stop_requested(g.img.skip)) return;  // device-driven solve: finished
  constexpr int TX = kAdjTX, TY = kAdjTY, NT = kAdjThreads;
  const ImgArgs &a = g.img;
  const int W = a.W, H = a.H, r = a.r, ntap = 4 * r + 1;
  const int AW = TX + 4 * r, AH = TY + 4 * r, GH = TY + 2 * r;
  double *red = reinterpret_cast<double *>(smem_raw);
  float *bufA = reinterpret_cast<float *>(smem_raw + 32 * sizeof(double));  // raw, AW x AH
  float *bufG = bufA + AW * AH;                                             // G_x raw, TX x GH
  float *bufM = bufG + TX * GH;                                             // M_x raw, TX x AH
  const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid / TX;
  const float alpha = a.alpha ? (float)(*a.alpha) : 0.f;
  const int n_work = LIST ? (int)(*a.tile_count) : 0;
  for (int wi = blockIdx.x, once = 1; LIST ? (wi < n_work) : (once != 0); wi += gridDim.x, once = 0) {
  const unsigned entry = LIST ? a.tile_list[wi] : (unsigned)wi;
  const int tile = (int)(entry & 0x3fffffffu);
  const int x0 = (tile % a.tiles_x) * TX, y0 = (tile / a.tiles_x) * TY;
  if (LIST) __syncthreads();  // LDS of the previous tile is free
  if (a.zero_ptr) {  // clear this tile of the other accumulation buffer (ping-pong: no memset launch next time)
    const bool dirty = LIST ? (entry & 0x40000000u) != 0 : (!a.flags_other || a.flags_other[tile] != 0);
    if (dirty) {
      for (int idx = tid; idx < TX * TY * a.zero_planes; idx += NT) {
        const int pl = idx / (TX * TY), q = idx - pl * (TX * TY);
        const int gx = x0 + (q % TX), gy = y0 + (q / TX);
        if (gx < W && gy < H) a.zero_ptr[(size_t)pl * W * H + (size_t)gy * W + gx] = 0.f;
      }
    }
    if (!LIST) {
      __syncthreads();
      if (tid == 0 && a.flags_other && dirty) a.flags_other[tile] = 0;
    }
  }
  const int slot = LIST ? wi : tile;
  if (LIST ? !(entry & 0x80000000u) : !tile_active(a, tile % a.tiles_x, tile / a.tiles_x, 2 * r, TX, TY)) {
    if (tid == 0) {
      a.partials[(size_t)0 * a.nblk + slot] = 0.0;
      a.partials[(size_t)1 * a.nblk + slot] = 0.0;
    }
    continue;
  }
  for (int idx = tid; idx < AW * AH; idx += NT) {
    const int ly = idx / AW, lx = idx - ly * AW;
    const int gx = reflect101(x0 + lx - 2 * r, W), gy = reflect101(y0 + ly - 2 * r, H);
    const size_t off = (size_t)gy * W + gx;
    float v = a.src_a[off];
    if (a.src_b) v = v + a.src_b[off];
    if (a.igp) v = a.igp[off] * alpha + v;
    bufA[idx] = v;
  }
  __syncthreads();
  {
    const bool interior = x0 >= 2 * r && x0 + TX - 1 <= W - 1 - 2 * r;
    const float *mrow = g.Mx + (size_t)(interior ? 2 * r : min(x0 + tx, W - 1)) * ntap;
    for (int ly = ty; ly < AH; ly += TY) {  // row pass, output column tx
      const float *S = bufA + ly * AW + tx;
      double m = 0.0;
      for (int i = 0; i < ntap; i++) m = __builtin_fma((double)mrow[i], (double)S[i], m);
      bufM[ly * TX + tx] = (float)m;
      if (ly >= r && ly < r + GH) {  // forward row pass, same op order as image_moments
        float sacc = a.taps[0] * S[r];
        for (int j = 1; j <= 2 * r; j++) sacc += a.taps[j] * S[r + j];
        bufG[(ly - r) * TX + tx] = sacc;
      }
    }
  }
  __syncthreads();
  double sI = 0, sII = 0;
  {
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
      const float *T = bufG + (ty + r) * TX + tx;
      float sacc = a.taps[r] * T[0];
      for (int t = 1; t <= r; t++) sacc += a.taps[r + t] * (T[t * TX] + T[-t * TX]);
      sI = (double)sacc;
      sII = (double)sacc * (double)sacc;
      if (a.out_blur0) a.out_blur0[(size_t)gy * W + gx] = sacc;
      const bool interior = y0 >= 2 * r && y0 + TY - 1 <= H - 1 - 2 * r;
      const float *mrow = g.My + (size_t)(interior ? 2 * r : min(y0 + ty, H - 1)) * ntap;
      const float *Q = bufM + ty * TX + tx;
      double j = 0.0;
      for (int i = 0; i < ntap; i++) j = __builtin_fma((double)mrow[i], (double)Q[i * TX], j);
      g.jt[(size_t)gy * W + gx] = (float)j;
    }
  }
  {
    double t0, t1;
    block_sum2(sI, sII, red, NT / 64, t0, t1);
    if (tid == 0) {
      a.partials[(size_t)0 * a.nblk + slot] = t0;
      a.partials[(size_t)1 * a.nblk + slot] = t1;
    }
  }
  }  // work loop
}

void launch_image_adjoint(const ImgAdjArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (a.Mx && a.My && a.img.r != 4 && a.img.r >= 1) {  // composite operator, any radius (tail finalize: r == 4 only)
    const size_t ldsg = image_adjoint2_lds_bytes(a.img.r);
    if (a.img.tile_list) CMX_LAUNCH((image_adjoint2g_kernel<true>), dim3(min(a.img.nblk, kTileListGrid)), dim3(kAdjThreads), ldsg, s, t0, t1, a);
    else CMX_LAUNCH((image_adjoint2g_kernel<false>), dim3(a.img.nblk), dim3(kAdjThreads), ldsg, s, t0, t1, a);
    return;
  }
  if (a.Mx && a.My && a.img.r == 4) {  // composite operator, radius 4: taps and coefficient rows in registers
    const size_t lds2 = image_adjoint2_lds_bytes(4);
    const bool tail = a.img.tail.counters != nullptr;
    if (a.img.tile_list) {
      const dim3 gl(min(a.img.nblk, kTileListGrid));
      if (tail) CMX_LAUNCH((image_adjoint2_kernel<4, true, 1>), gl, dim3(kAdjThreads), lds2, s, t0, t1, a);
      else CMX_LAUNCH((image_adjoint2_kernel<4, true, 0>), gl, dim3(kAdjThreads), lds2, s, t0, t1, a);
    } else {
      if (a.img.macc) CMX_LAUNCH((image_adjoint2_kernel<4, false, 3>), dim3(a.img.nblk), dim3(kAdjThreads), lds2, s, t0, t1, a);
      else if (tail && a.img.tail.fin.chain.sm) CMX_LAUNCH((image_adjoint2_kernel<4, false, 2>), dim3(a.img.nblk), dim3(kAdjThreads), lds2, s, t0, t1, a);
      else if (tail) CMX_LAUNCH((image_adjoint2_kernel<4, false, 1>), dim3(a.img.nblk), dim3(kAdjThreads), lds2, s, t0, t1, a);
      else CMX_LAUNCH((image_adjoint2_kernel<4, false, 0>), dim3(a.img.nblk), dim3(kAdjThreads), lds2, s, t0, t1, a);
    }
    return;
  }
  const size_t lds = image_adjoint_lds_bytes(a.img.r);
  if (a.img.tile_list) {
    const dim3 g(min(a.img.nblk, kTileListGrid));
    if (a.img.r == 4) CMX_LAUNCH((image_adjoint_kernel<4, kAdjTX, kAdjTY, kAdjThreads, true>), g, dim3(kAdjThreads), lds, s, t0, t1, a);
    else CMX_LAUNCH((image_adjoint_kernel<-1, kAdjTX, kAdjTY, kAdjThreads, true>), g, dim3(kAdjThreads), lds, s, t0, t1, a);
  } else {
    const dim3 g(a.img.nblk);
    if (a.img.r == 4) CMX_LAUNCH((image_adjoint_kernel<4, kAdjTX, kAdjTY, kAdjThreads, false>), g, dim3(kAdjThreads), lds, s, t0, t1, a);
    else CMX_LAUNCH((image_adjoint_kernel<-1, kAdjTX, kAdjTY, kAdjThreads, false>), g, dim3(kAdjThreads), lds, s, t0, t1, a);
  }
}

// ---------------------------------------------------------------------------------------------- gather passes
// d(contrast)/d(theta_k) = (2/N) sum_events [ r0_k * dJt/dx(at the event) + r1_k * dJt/dy ] where the
// bilinear-interpolation derivatives are exactly the signed-weight sums the reference scatters into its derivative
// images (local_image_warped_events.cpp:163-166, event_pano_warper.cpp:327-330).
__device__ __forceinline__ void bilinear_grad(const float *it, int W, int xx, int yy, float dx, float dy, float &A, float &B) {
  const float *q = it + (size_t)yy * W + xx;
  const float i00 = q[0], i01 = q[1], i10 = q[W], i11 = q[W + 1];
  A = (1.f - dy) * (i01 - i00) + dy * (i11 - i10);
  B = (1.f - dx) * (i10 - i00) + dx * (i11 - i01);
}

// the same two directional derivatives of the separable plane c(x,y) = cx[x]*cy[y] (= G^T 1); zero away from the border
__device__ __forceinline__ void border_grad(const float *cx, const float *cy, int W, int H, int r, int xx, int yy, float dx,
                                            float dy, float &A, float &B) {
  A = 0.f;
  B = 0.f;
  if (xx <= r || xx + 1 >= W - 1 - r || yy <= r || yy + 1 >= H - 1 - r) {
    const float c00 = cx[xx] * cy[yy], c01 = cx[xx + 1] * cy[yy], c10 = cx[xx] * cy[yy + 1], c11 = cx[xx + 1] * cy[yy + 1];
    A = (1.f - dy) * (c01 - c00) + dy * (c11 - c10);
    B = (1.f - dx) * (c10 - c00) + dx * (c11 - c01);
  }
}

int gather_blocks(int n) {
  int blocks = (n + 255) / 256;
  const int cap = 768;  // 3 workgroups per CU (be_gather is fp64-ALU bound at ~150 VGPRs: 3 blocks/CU is its occupancy)
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}
// front end: one workgroup per kFeGatherPerBlock events up to the cap.  Swept on MI355X with the bearing / dt streams
// (tools/sweep_fe_gather2.sh, two events in flight per thread): the kernel alone is indifferent between 512 and 1024
// events (10.9-11.3 us), 2048 -> 12.2 us, 3072 -> 16 us; the evaluation prefers 1024 (47.2 us against 49.2 us at 512)
// because finalize then sums half as many partial rows
// (round 5, profiles/r05_fe_shape.txt: threads per workgroup x events per workgroup swept with the tail finalize on -- the two
//  macros exist for that sweep, tools/sweep_fe_shape.sh)
#ifndef CMX_FE_GATHER_NT
#define CMX_FE_GATHER_NT 256
#endif
#ifndef CMX_FE_GATHER_PER_BLOCK
#define CMX_FE_GATHER_PER_BLOCK 1024
#endif
constexpr int kFeGatherNT = CMX_FE_GATHER_NT;
constexpr int kFeGatherPerBlock = CMX_FE_GATHER_PER_BLOCK, kFeGatherCap = 2048;
int fe_gather_blocks(int n) {
  // small packets (the reference's own: tens of thousands of events) would leave most CUs without a workgroup at 1024 events
  // each: below 512k events the slices shrink to n / 512 events (a multiple of 256, at least 256) -- 60k events: 235 workgroups
  // instead of 59, a device-driven solve 0.86 -> 0.79-0.82 ms (same-box A/B of builds); at 1M events 1024 stays (gather 13.5 us
  // against 15.0 at 512 and 15.2 at 768)
  int per = kFeGatherPerBlock;
  if (n < 512 * 1024) {
    per = ((n / 512 + kFeGatherNT - 1) / kFeGatherNT) * kFeGatherNT;
    per = per < kFeGatherNT ? kFeGatherNT : (per > kFeGatherPerBlock ? kFeGatherPerBlock : per);
  }
  int blocks = (n + per - 1) / per;
  return blocks < 1 ? 1 : (blocks > kFeGatherCap ? kFeGatherCap : blocks);
}

// The front-end gather's event loop for the tile-ordered streams (the production form), in branch-free phases: every load of a
// phase is in flight at once -- U x 24 B of streams, then (after the U warps) the U x 4 Jt cells.  With the loads behind
// `if (act && ok)` the compiler cannot hoist them and the events' trips to the Jt plane follow one another.  Same operations per
// event, same order of a thread's additions: the sums are the generic loop's bit for bit.  At 1M events the launch is bound by its
// fixed costs and the two forms time the same (profiles/r04_fe_gather_anatomy.txt); at 16M events per launch the loop IS the launch
// (116 -> 99 us).
template <int U>
__device__ __forceinline__ void fe_gather_streams(const FeGatherArgs &g, const FeSplatArgs &a, int blk_beg, int blk_end, double acc[3],
                                                  double acc2[3]) {
  const int W = a.W, H = a.H, r = g.r;
  for (int i0 = blk_beg + (int)threadIdx.x; i0 < blk_end; i0 += kFeGatherNT * U) {
    double2 bv[U];
    double dt[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * kFeGatherNT;
      ok[u] = i < blk_end;
      const int ii = ok[u] ? i : blk_beg;
      bv[u] = *reinterpret_cast<const double2 *>(g.sb + 2 * (size_t)ii);
      dt[u] = g.sdt[ii];
    }
    FeWarp w[U];
    const float *q[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      w[u] = fe_warp_math<true>(a, bv[u].x, bv[u].y, 1.0, dt[u]);
      ok[u] = ok[u] && w[u].ok;
      q[u] = g.itilde + (ok[u] ? (size_t)w[u].yy * W + w[u].xx : (size_t)0);  // (cells 0 .. W+1 exist in every image the path accepts)
    }
    float i00[U], i01[U], i10[U], i11[U];
#pragma unroll
    for (int u = 0; u < U; u++) { i00[u] = q[u][0]; i01[u] = q[u][1]; i10[u] = q[u][W]; i11[u] = q[u][W + 1]; }
    bool edge = false;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float dx = w[u].dx, dy = w[u].dy;
      const float A = (1.f - dy) * (i01[u] - i00[u]) + dy * (i11[u] - i10[u]);
      const float B = (1.f - dx) * (i10[u] - i00[u]) + dx * (i11[u] - i01[u]);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double t = (double)w[u].r0[k] * (double)A + (double)w[u].r1[k] * (double)B;
        acc[k] = ok[u] ? acc[k] + t : acc[k];
      }
      edge = edge || (ok[u] && (w[u].xx <= r || w[u].xx + 1 >= W - 1 - r || w[u].yy <= r || w[u].yy + 1 >= H - 1 - r));
    }
    if (g.cx && __any(edge)) {  // votes within r of the border: rare, wave-uniform
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!ok[u]) continue;
        float Ac, Bc;
        border_grad(g.cx, g.cy, W, H, r, w[u].xx, w[u].yy, w[u].dx, w[u].dy, Ac, Bc);
        if (Ac != 0.f || Bc != 0.f) {
#pragma unroll
          for (int k = 0; k < 3; k++) acc2[k] += (double)w[u].r0[k] * (double)Ac + (double)w[u].r1[k] * (double)Bc;
        }
      }
    }
  }
}
// U = 4 (same-box A/B of builds, gather us at 1M / 4M / 16M events): generic loop 13.6 / 29.0 / 116; phases U = 2 13.6 / 30.4 / 110;
// phases U = 4 13.8 / 29.4 / 99 (127 VGPRs: four waves per SIMD, what ~1000 workgroups of four waves need)
#ifndef CMX_FE_GATHER_U
#define CMX_FE_GATHER_U 4
#endif

// CHAIN: 0 plain; 1 device-driven solve, gated by the flag the cost stage's finalize wrote; 2 device-driven solve, SELF-GATING:
// the image pass in front left the image's two moments in accumulator rows and ran no finalize -- every workgroup of this
// launch forms the contrast from them (16 loads, the same expression the finalize uses) and evaluates the machine's
// acceptance test itself.  Test passed: the launch is the gradient pass, its last-arriving workgroup runs the finalize and
// BOTH machine steps (cost, then gradient).  Test failed: workgroup 0 alone runs the cost finalize and the machine's step, the
// others leave.  The image pass loses its tail (last-arriver protocol + finalize + step: ~4 us of its 13) at every point.
template <int CHAIN>
__global__ __launch_bounds__(kFeGatherNT) void fe_gather_kernel(FeGatherArgs g) {
  if (CHAIN != 2 && g.gate && *g.gate == 0) return;  // gated gradient pass: the cost-only evaluation in front decided against it
  if (CHAIN && wg_stop_requested(g.ev.skip)) return;  // device-driven solve: finished
  __shared__ double red[(kFeGatherNT / 64) * 6];
  __shared__ FinSmem fin_sm;
  if (CHAIN == 2) {
    const FinalizeArgs &fa = g.tail.fin;
    double s0, s1, mu;
    chain_moment_sums(fa.macc, s0, s1);
    const double c = contrast_from_sums(s0, s1, fa.npix, fa.measure, &mu);
    // (first slot of a warm start: the machine's first request is cost + gradient, gate mode 4 -- its state is still on the host)
    // the gate is read from the slot-parity copy the PREVIOUS slot's finalize published: this launch's own finalize may already
    // be rewriting the machine (workgroup 0, cost-only outcome) while late workgroups arrive here
    if (!fa.chain.sm_src && !gate_condition(c, fa.chain.gate_cur->thr, fa.chain.gate_cur->mode)) {
      if (blockIdx.x != 0) return;
      FinalizeArgs f = fa;  // cost only: no gradient sums to read
      f.gP = 0;
      f.gacc = nullptr;
      finalize_body<kFeGatherNT, true>(f, fin_sm);
      return;
    }
  }
  if (CHAIN) fe_resolve_omega(g.ev);
  const FeSplatArgs &a = g.ev;
  double acc[3] = {0, 0, 0}, acc2[3] = {0, 0, 0};
  constexpr int U = 2;  // events in flight per thread (swept on MI355X: 2 -> 11.9 us, 1 -> 12.2, 4 -> 12.9, 8 -> 14.9 per 1M events)
  // every workgroup walks ONE contiguous slice of the event list (in tile order that keeps its LUT / Itilde reads local)
  const int per_block = ((a.n + (int)gridDim.x - 1) / (int)gridDim.x + kFeGatherNT - 1) / kFeGatherNT * kFeGatherNT;
  const int blk_beg = blockIdx.x * per_block, blk_end = min(a.n, blk_beg + per_block);
  const int stride = kFeGatherNT;
  // (the device-driven solve's variants keep the generic loop: with the finalize and the machine's step inlined, four events in
  //  flight cost them registers -- a 1M-event solve 0.558 -> 0.585 ms, same-box A/B -- and their launches are never large)
  if (CHAIN == 0 && CMX_FE_GATHER_U > 0 && g.sb) fe_gather_streams<(CMX_FE_GATHER_U > 0 ? CMX_FE_GATHER_U : 1)>(g, a, blk_beg, blk_end, acc, acc2);
  else
  for (int i0 = blk_beg + threadIdx.x; i0 < blk_end; i0 += stride * U) {
    double dt[U], px[U], py[U], pz[U];
    bool act[U];
    if (g.sb) {  // tile order with per-event bearing / dt streams (coalesced): the only gathers left are the Jt reads
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = i0 + u * stride;
        act[u] = i < blk_end;
        const int ii = act[u] ? i : blk_beg;
        const double2 v = *reinterpret_cast<const double2 *>(g.sb + 2 * (size_t)ii);
        px[u] = v.x; py[u] = v.y; pz[u] = 1.0;
        dt[u] = g.sdt[ii];
      }
    } else {
      if (g.tb && !g.sxy) {  // time order with the bearing stream (deterministic mode)
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int i = i0 + u * stride;
          act[u] = i < blk_end;
          const int ii = act[u] ? i : blk_beg;
          const double2 v = *reinterpret_cast<const double2 *>(g.tb + 2 * (size_t)ii);
          px[u] = v.x; py[u] = v.y; pz[u] = 1.0;
          dt[u] = a.batch_dt[ii / a.per_batch];
        }
      } else {
      uint32_t e[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = i0 + u * stride;
        act[u] = i < blk_end;
        if (g.sxy) {
          e[u] = act[u] ? g.sxy[i] : 0u;
          dt[u] = act[u] ? a.batch_dt[g.sbatch[i]] : 0.0;
        } else {
          e[u] = act[u] ? a.xy[i] : 0u;
          dt[u] = act[u] ? a.batch_dt[i / a.per_batch] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        load_bearing(a, (int)(e[u] & 0xffff), (int)((e[u] >> 16) & 0x7fff), px[u], py[u], pz[u]);
      }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const FeWarp w = fe_warp_math<true>(a, px[u], py[u], pz[u], dt[u]);
      if (act[u] && w.ok) {
        float A, B;
        bilinear_grad(g.itilde, a.W, w.xx, w.yy, w.dx, w.dy, A, B);
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] += (double)w.r0[k] * (double)A + (double)w.r1[k] * (double)B;
        if (g.cx) {
          float Ac, Bc;
          border_grad(g.cx, g.cy, a.W, a.H, g.r, w.xx, w.yy, w.dx, w.dy, Ac, Bc);
          if (Ac != 0.f || Bc != 0.f) {
#pragma unroll
            for (int k = 0; k < 3; k++) acc2[k] += (double)w.r0[k] * (double)Ac + (double)w.r1[k] * (double)Bc;
          }
        }
      }
    }
  }
  // partial table layout: [column][block], columns = S1 (3) then S2 (3).  One barrier for all six sums: wave sums,
  // one LDS row per wave, six threads add the four rows (in the order block_sum uses).  The border sums S2 are zero
  // for almost every wave: their shuffles are skipped wave-uniformly.
  double v[6];
#pragma unroll
  for (int k = 0; k < 3; k++) v[k] = wave_sum(acc[k]);
  const bool any2 = g.cx && __any(acc2[0] != 0.0 || acc2[1] != 0.0 || acc2[2] != 0.0);
#pragma unroll
  for (int k = 0; k < 3; k++) v[3 + k] = any2 ? wave_sum(acc2[k]) : 0.0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) red[wave * 6 + k] = v[k];
  }
  __syncthreads();
  const bool tail = g.tail.counters != nullptr;
  if (threadIdx.x < (g.cx ? 6 : 3)) {
    const int k = threadIdx.x;
    double v = red[k] + red[6 + k] + red[12 + k] + red[18 + k];
#pragma unroll
    for (int wv = 4; wv < kFeGatherNT / 64; wv++) v += red[6 * wv + k];
    if (g.tail.fin.gacc) {  // accumulator rows instead of the table (see FinalizeArgs::gacc); with or without the tail
      if (v != 0.0)
        __hip_atomic_fetch_add(g.tail.fin.gacc + (size_t)(blockIdx.x % kTailShards) * g.tail.fin.gacc_stride + k, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    } else if (tail) st_sc1(g.gpartials + (size_t)k * gridDim.x + blockIdx.x, v);  // write-through: read by the last arriver
    else g.gpartials[(size_t)k * gridDim.x + blockIdx.x] = v;
  }
  if (CHAIN == 0 && tail && g.tail.poll) {
    // Polling tail: the sums above are on their way as agent-scope atomics; drain them, then everybody but workgroup 0 ARRIVES
    // (one fire-and-forget atomic) and leaves.  Workgroup 0 -- dispatched first, long done with its own slice -- polls the count with
    // one lane and finalizes.  It waits for workgroups that wait for nobody; the wait is bounded all the same (1 s: only a dead device
    // gets there; the result then carries kFuseIncomplete in its fallback word).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the arrivals are sharded over the kTailShards + 1 counter lines: ~1000 atomics on ONE memory-side address serialise at ~12 ns
    //  each -- the single-counter form of this tail took the launch from 13.9 to 21.6 us)
    constexpr int kLines = kTailShards + 1;
    if (blockIdx.x != 0) {
      if (threadIdx.x == 0)
        __hip_atomic_fetch_add(g.tail.counters + (blockIdx.x % kLines) * kTailStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (threadIdx.x < 64) {  // wave 0: lane q < kLines polls line q
      const unsigned want = gridDim.x - 1u;
      const unsigned long long t0 = wall_clock64();
      const int q = threadIdx.x;
      for (;;) {
        unsigned v = q < kLines ? __hip_atomic_load(g.tail.counters + q * kTailStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);  // (lanes 0..15 hold the sum of the lines)
        const unsigned total = __shfl(v, 0, 64);
        if (total >= want) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 100000000ull) {
          if (q == 0 && g.tail.fin.fallback) atomicOr(g.tail.fin.fallback, kFuseIncomplete);
          break;
        }
      }
      if (q < kLines) __hip_atomic_store(g.tail.counters + q * kTailStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    finalize_body<kFeGatherNT, false>(g.tail.fin, fin_sm);
    return;
  }
  if (tail && tail_arrive(g.tail, (int)gridDim.x, (int)blockIdx.x, fin_sm)) finalize_body<kFeGatherNT, CHAIN != 0>(g.tail.fin, fin_sm);
}

int launch_fe_gather(const FeGatherArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  const int blocks = fe_gather_blocks(a.ev.n);
  if (a.tail.fin.chain.sm && a.tail.fin.chain.stage == 2) CMX_LAUNCH(fe_gather_kernel<2>, dim3(blocks), dim3(kFeGatherNT), 0, s, t0, t1, a);
  else if (a.tail.fin.chain.sm || a.ev.w_dev) CMX_LAUNCH(fe_gather_kernel<1>, dim3(blocks), dim3(kFeGatherNT), 0, s, t0, t1, a);
  else CMX_LAUNCH(fe_gather_kernel<0>, dim3(blocks), dim3(kFeGatherNT), 0, s, t0, t1, a);
  return blocks;
}

// back end: per event V = (dItilde/dx, dItilde/dy) * dpm_ddrot (3-vector); events of one batch share the 3x3N
// spline Jacobian, so V is summed per batch first (segmented wave reduction over the contiguous batch runs, then
// LDS), and one small mat-vec per batch maps it onto the 3N knot parameters the batch touches.
constexpr int kMaxGradLDS = 3 * kMaxKnots;

// Pass 1: every WAVE walks its own 64-event slices (time order, so the events of a batch are adjacent lanes): segmented
// shuffle reduction of V (and of the border-term vector U) over the batch runs inside the wave; the run's head lane
// stores the partial sums to vparts[batch][part] (part = slice index relative to the batch's first slice: a plain
// store, no atomics, every slot written exactly once).  No barrier, no LDS.
__global__ __launch_bounds__(256) void be_gather_kernel(BeGatherArgs g) {
  const BeSplatArgs &a = g.ev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  for (int base = (blockIdx.x * 4 + wave) * 64; base < a.n; base += nwaves * 64) {
    const int i = base + lane;
    double V0 = 0, V1 = 0, V2 = 0, U0 = 0, U1 = 0, U2 = 0;
    int batch = -1;
    if (i < a.n) {
      const BeWarp w = be_warp_event<2>(a, i);
      batch = w.batch;
      if (w.ok) {
        float A, B;
        bilinear_grad(g.itilde, a.Wp, w.xx, w.yy, w.dx, w.dy, A, B);
        V0 = (double)A * (double)w.m[0] + (double)B * (double)w.m[3];
        V1 = (double)A * (double)w.m[1] + (double)B * (double)w.m[4];
        V2 = (double)A * (double)w.m[2] + (double)B * (double)w.m[5];
        float Ac, Bc;
        border_grad(g.cx, g.cy, a.Wp, a.Hp, g.r, w.xx, w.yy, w.dx, w.dy, Ac, Bc);
        if (Ac != 0.f || Bc != 0.f) {  // rare: votes within r of the panorama border
          U0 = (double)Ac * (double)w.m[0] + (double)Bc * (double)w.m[3];
          U1 = (double)Ac * (double)w.m[1] + (double)Bc * (double)w.m[4];
          U2 = (double)Ac * (double)w.m[2] + (double)Bc * (double)w.m[5];
        }
      }
    }
    const bool any_u = __any(U0 != 0.0 || U1 != 0.0 || U2 != 0.0);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t0 = __shfl_down(V0, o, 64), t1 = __shfl_down(V1, o, 64), t2 = __shfl_down(V2, o, 64);
      const int bo = __shfl_down(batch, o, 64);
      const bool same = lane + o < 64 && bo == batch;
      if (same) { V0 += t0; V1 += t1; V2 += t2; }
      if (any_u) {  // wave-uniform
        const double u0 = __shfl_down(U0, o, 64), u1 = __shfl_down(U1, o, 64), u2 = __shfl_down(U2, o, 64);
        if (same) { U0 += u0; U1 += u1; U2 += u2; }
      }
    }
    const int bprev = __shfl_up(batch, 1, 64);
    if (batch >= 0 && (lane == 0 || bprev != batch)) {
      const int part = (base >> g.slice_shift) - ((batch * a.per_batch) >> g.slice_shift);
      double *dst = g.vparts + ((size_t)batch * g.parts_per_batch + part) * 6;
      dst[0] = V0; dst[1] = V1; dst[2] = V2;
      dst[3] = U0; dst[4] = U1; dst[5] = U2;
    }
  }
}

// The same pass with FOUR consecutive events per lane (per_batch % 4 == 0, so a lane's events share one batch): one
// 16-byte event load and one rotation-table read per lane, the four warps are independent instruction streams, and
// the segmented wave reduction -- a third of the one-event form's instructions -- is paid once per 256 events.
// NF = 0: per-batch partial sums go to vparts and be_gather_batch_kernel finishes them.  NF = 2 / 4 (spline order): the
// per-batch pass is FOLDED in -- the head lane of every batch run applies the batch's 3 x 3NF Jacobian to its partial V
// (J is linear: partial sums may be multiplied one by one) and adds the columns to workgroup accumulators in LDS; the
// workgroup's sums go to the accumulator rows of the tail finalize (FinalizeArgs::gacc) and the last-arriving workgroup
// finalizes.  No vparts round trip, no per-batch launch.  Not in CMX_OPT_DETERMINISTIC (order of the fp64 atomics).
template <int NF>
__global__ __launch_bounds__(256) void be_gather4_kernel(BeGatherArgs g) {
  __shared__ double shG[NF ? kMaxGradLDS : 1], shG2[NF ? kMaxGradLDS : 1];
  __shared__ FinSmem fin_sm;
  if (g.gate && *g.gate == 0) return;  // gated gradient pass (see fe_gather_kernel)
  if (NF) {
    for (int j = threadIdx.x; j < g.P; j += 256) { shG[j] = 0; shG2[j] = 0; }
    __syncthreads();
  }
  const BeSplatArgs &a = g.ev;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
#ifdef CMX_BE_GATHER_LDS_POSE
  // EXPERIMENT (measured, rejected: profiles/r06_be_lds_pose.txt): the rotations of the batches a wave pass meets
  // (<= kPoseRuns) fetched ONE PASS AHEAD into a per-wave LDS slot by 9 x kPoseRuns lanes -- one 72-byte load per batch run instead of
  // one per lane, off the pass's critical path, no registers held across the warp -- and read back from LDS by every lane.
  constexpr int kPoseRuns = 4;  // 256 events of a pass span at most 256 / per_batch + 2 batches; the launcher's per_batch >= 100
  __shared__ double shR[2][4][kPoseRuns * 9];
  auto load_pose = [&](int base_next) -> double {  // (the value stays in ONE register pair across the pass: the LDS write comes at its end)
    if (lane < kPoseRuns * 9 && base_next < a.n) {
      const int b = min(base_next / a.per_batch + lane / 9, (a.n - 1) / a.per_batch);
      return a.poseR[b].R[lane % 9];
    }
    return 0.0;
  };
  int pslot = 0;
  if (lane < kPoseRuns * 9) shR[0][wave][lane] = load_pose((blockIdx.x * 4 + wave) * 256);
#endif
  for (int base = (blockIdx.x * 4 + wave) * 256; base < a.n; base += nwaves * 256) {
    const int i0 = base + lane * 4;
    double V0 = 0, V1 = 0, V2 = 0, U0 = 0, U1 = 0, U2 = 0;
    int batch = -1;
#ifdef CMX_BE_GATHER_LDS_POSE
    const double rnext = load_pose(base + nwaves * 256);  // the NEXT pass's rotations: in flight while this pass warps
#endif
    if (i0 < a.n) {
      batch = i0 / a.per_batch;
      uint32_t e[4];
      if (i0 + 3 < a.n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(a.xy + i0);
        e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = (i0 + u < a.n) ? a.xy[i0 + u] : 0u;
      }
      double R[9];
#ifdef CMX_BE_GATHER_LDS_POSE
      {
        const double *Rl = shR[pslot][wave] + (batch - base / a.per_batch) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = Rl[k];
      }
#else
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = a.poseR[batch].R[k];
#endif
      double rx[4], ry[4], rz[4];  // e_ray_w = R * bearing of the lane's four events
      if (g.tb && i0 + 3 < a.n) {  // 64 contiguous bytes per lane from the time-ordered bearing stream: (x, y), z = 1
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const double2 v = *reinterpret_cast<const double2 *>(g.tb + 2 * (size_t)(i0 + u));
          be_rotate<true>(R, v.x, v.y, 1.0, rx[u], ry[u], rz[u]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          double b0, b1, b2;
          load_bearing(a, (int)(e[u] & 0xffff), (int)((e[u] >> 16) & 0x7fff), b0, b1, b2);
          be_rotate<false>(R, b0, b1, b2, rx[u], ry[u], rz[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (i0 + u >= a.n) break;
        const BeWarp w = be_project<2>(a, e[u], batch, rx[u], ry[u], rz[u]);
        if (w.ok) {
#pragma clang fp contract(fast)  // the fp64 sums of fp32 x fp32 products (exact in fp64): two FMAs per component instead of two multiplies + two adds
          float A, B;
          bilinear_grad(g.itilde, a.Wp, w.xx, w.yy, w.dx, w.dy, A, B);
          const double Ad = (double)A, Bd = (double)B;
          V0 = __builtin_fma(Bd, (double)w.m[3], __builtin_fma(Ad, (double)w.m[0], V0));
          V1 = __builtin_fma(Bd, (double)w.m[4], __builtin_fma(Ad, (double)w.m[1], V1));
          V2 = __builtin_fma(Bd, (double)w.m[5], __builtin_fma(Ad, (double)w.m[2], V2));
          float Ac, Bc;
          border_grad(g.cx, g.cy, a.Wp, a.Hp, g.r, w.xx, w.yy, w.dx, w.dy, Ac, Bc);
          if (Ac != 0.f || Bc != 0.f) {  // rare: votes within r of the panorama border
            U0 += (double)Ac * (double)w.m[0] + (double)Bc * (double)w.m[3];
            U1 += (double)Ac * (double)w.m[1] + (double)Bc * (double)w.m[4];
            U2 += (double)Ac * (double)w.m[2] + (double)Bc * (double)w.m[5];
          }
        }
      }
    }
#ifdef CMX_BE_GATHER_LDS_POSE
    pslot ^= 1;
    if (lane < kPoseRuns * 9) shR[pslot][wave][lane] = rnext;
#endif
    const bool any_u = __any(U0 != 0.0 || U1 != 0.0 || U2 != 0.0);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t0 = __shfl_down(V0, o, 64), t1 = __shfl_down(V1, o, 64), t2 = __shfl_down(V2, o, 64);
      const int bo = __shfl_down(batch, o, 64);
      const bool same = lane + o < 64 && bo == batch;
      if (same) { V0 += t0; V1 += t1; V2 += t2; }
      if (any_u) {  // wave-uniform
        const double u0 = __shfl_down(U0, o, 64), u1 = __shfl_down(U1, o, 64), u2 = __shfl_down(U2, o, 64);
        if (same) { U0 += u0; U1 += u1; U2 += u2; }
      }
    }
    if (NF) {
      // lane = (run r of this wave pass, column c of the run's batch Jacobian): a wave pass of 256 events meets at most
      // 256 / per_batch + 2 <= 64 / (3 NF) batch runs (the launcher's condition).  The first lane of a run holds the run's
      // sums: every (run, column) lane fetches them and adds its column.  (Fetching the Jacobian entries a pass ahead
      // kept six more registers alive across the warp: 172 VGPRs, two waves per SIMD instead of three, 60 -> 78 us.)
      const int fr = lane / (3 * NF), fc = lane - fr * (3 * NF);
      const int fb = base / a.per_batch + fr;
      const int first_ev = max(base, fb * a.per_batch);
      const bool f_ok = first_ev < min(a.n, base + 256);
      const int fhead = (first_ev - base) >> 2;
      float fj0 = 0.f, fj1 = 0.f, fj2 = 0.f;
      int fj = -1;
      if (f_ok) {
        const PoseEntry &pe = a.poses[fb];
        fj0 = pe.Jcp[fc]; fj1 = pe.Jcp[3 * NF + fc]; fj2 = pe.Jcp[6 * NF + fc];
        fj = 3 * (pe.idx_cp_beg - a.num_fixed) + fc;
      }
      const double v0 = __shfl(V0, fhead, 64), v1 = __shfl(V1, fhead, 64), v2 = __shfl(V2, fhead, 64);
      if (f_ok && fj >= 0) atomicAdd(&shG[fj], v0 * (double)fj0 + v1 * (double)fj1 + v2 * (double)fj2);
      if (any_u) {  // wave-uniform
        const double u0 = __shfl(U0, fhead, 64), u1 = __shfl(U1, fhead, 64), u2 = __shfl(U2, fhead, 64);
        if (f_ok && fj >= 0 && (u0 != 0.0 || u1 != 0.0 || u2 != 0.0))
          atomicAdd(&shG2[fj], u0 * (double)fj0 + u1 * (double)fj1 + u2 * (double)fj2);
      }
    } else {
      const int bprev = __shfl_up(batch, 1, 64);
      if (batch >= 0 && (lane == 0 || bprev != batch)) {
        const int part = (base >> 8) - ((batch * a.per_batch) >> 8);
        double *dst = g.vparts + ((size_t)batch * g.parts_per_batch + part) * 6;
        dst[0] = V0; dst[1] = V1; dst[2] = V2;
        dst[3] = U0; dst[4] = U1; dst[5] = U2;
      }
    }
  }
  if (NF) {
    __syncthreads();
    double *row = g.tail.fin.gacc + (size_t)(blockIdx.x % kTailShards) * g.tail.fin.gacc_stride;
    for (int j = threadIdx.x; j < g.P; j += 256) {
      const double v1 = shG[j], v2 = shG2[j];
      if (v1 != 0.0) __hip_atomic_fetch_add(row + j, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v2 != 0.0) __hip_atomic_fetch_add(row + g.P + j, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (g.tail.counters && tail_arrive(g.tail, (int)gridDim.x, (int)blockIdx.x, fin_sm)) finalize_body<256>(g.tail.fin, fin_sm);
  }
}

// Pass 2: one thread per batch: sum its parts, apply the batch's 3x3N spline Jacobian (events of one batch share it),
// accumulate S1 / S2 per parameter in LDS, one partial row per workgroup ([column][block]).
template <int N, bool DET>
__global__ __launch_bounds__(256) void be_gather_batch_kernel(BeGatherArgs g, int nb) {
  __shared__ double shG[kMaxGradLDS], shG2[kMaxGradLDS];
  __shared__ double red[8];
  __shared__ FinSmem fin_sm;
  const BeSplatArgs &a = g.ev;
  const int tid = threadIdx.x;
  for (int j = tid; j < g.P; j += 256) { shG[j] = 0; shG2[j] = 0; }
  __syncthreads();
  for (int b0 = blockIdx.x * 256; b0 < nb; b0 += gridDim.x * 256) {  // uniform trip count: the DET form has barriers
    const int b = b0 + tid;
    double V[6] = {0, 0, 0, 0, 0, 0};
    int jbase = -(1 << 28);
    double c1[3 * N], c2[3 * N];
#pragma unroll
    for (int c = 0; c < 3 * N; c++) { c1[c] = 0; c2[c] = 0; }
    bool has_u = false;
    if (b < nb) {
      const int first = b * a.per_batch, last = min(a.n, first + a.per_batch) - 1;
      const int nparts = (last >> g.slice_shift) - (first >> g.slice_shift) + 1;
      for (int p = 0; p < nparts; p++) {
        const double *src = g.vparts + ((size_t)b * g.parts_per_batch + p) * 6;
#pragma unroll
        for (int q = 0; q < 6; q++) V[q] += src[q];
      }
      const PoseEntry &pe = a.poses[b];
      jbase = 3 * (pe.idx_cp_beg - a.num_fixed);
      has_u = V[3] != 0.0 || V[4] != 0.0 || V[5] != 0.0;
#pragma unroll
      for (int c = 0; c < 3 * N; c++) {
        const double j0 = (double)pe.Jcp[c], j1 = (double)pe.Jcp[3 * N + c], j2 = (double)pe.Jcp[6 * N + c];
        c1[c] = V[0] * j0 + V[1] * j1 + V[2] * j2;
        if (has_u) c2[c] = V[3] * j0 + V[4] * j1 + V[5] * j2;
      }
    }
    if (!DET) {
#pragma unroll
      for (int c = 0; c < 3 * N; c++) {
        const int j = jbase + c;
        if (j >= 0) {
          atomicAdd(&shG[j], c1[c]);
          if (has_u) atomicAdd(&shG2[j], c2[c]);
        }
      }
    } else {
      // deterministic mode: one block-wide sum per parameter, lanes / waves / batches always in the same order
      const bool any_u = __syncthreads_or(has_u ? 1 : 0) != 0;
      for (int j = 0; j < g.P; j++) {
        double v1 = 0, v2 = 0;
#pragma unroll
        for (int c = 0; c < 3 * N; c++)
          if (jbase + c == j) { v1 = c1[c]; v2 = c2[c]; }
        double t1, t2 = 0;
        if (any_u) block_sum2(v1, v2, red, 4, t1, t2);
        else t1 = block_sum(v1, red);
        if (tid == 0) { shG[j] += t1; shG2[j] += t2; }
      }
    }
  }
  __syncthreads();
  const bool tail = g.tail.counters != nullptr;
  if (g.tail.fin.gacc) {  // accumulator rows instead of the table (see FinalizeArgs::gacc); with or without the tail
    double *row = g.tail.fin.gacc + (size_t)(blockIdx.x % kTailShards) * g.tail.fin.gacc_stride;
    for (int j = tid; j < g.P; j += 256) {
      const double v1 = shG[j], v2 = shG2[j];
      if (v1 != 0.0) __hip_atomic_fetch_add(row + j, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v2 != 0.0) __hip_atomic_fetch_add(row + g.P + j, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else
  for (int j = tid; j < g.P; j += 256) {  // [column][block]
    if (tail) {  // write-through: read by the last-arriving workgroup of this launch
      st_sc1(g.gpartials + (size_t)j * gridDim.x + blockIdx.x, shG[j]);
      st_sc1(g.gpartials + (size_t)(g.P + j) * gridDim.x + blockIdx.x, shG2[j]);
    } else {
      g.gpartials[(size_t)j * gridDim.x + blockIdx.x] = shG[j];
      g.gpartials[(size_t)(g.P + j) * gridDim.x + blockIdx.x] = shG2[j];
    }
  }
  if (tail && tail_arrive(g.tail, (int)gridDim.x, (int)blockIdx.x, fin_sm)) finalize_body<256>(g.tail.fin, fin_sm);
}

__global__ __launch_bounds__(256) void bearing_stream_kernel(const uint32_t *xy, const double *lut2, int W, int n, double *tb) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t e = xy[i];
    *reinterpret_cast<double2 *>(tb + 2 * (size_t)i) =
        *reinterpret_cast<const double2 *>(lut2 + 2 * ((size_t)((e >> 16) & 0x7fff) * W + (e & 0xffff)));
  }
}
void launch_bearing_stream(const uint32_t *xy, const double *lut2, int W, int n, double *tb, hipStream_t s) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bearing_stream_kernel, dim3(blocks), dim3(256), 0, s, xy, lut2, W, n, tb);
}

int be_batch_blocks(int nb) {
  const int blocks = (nb + 255) / 256;
  return blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
}

bool be_gather_folds(const BeGatherArgs &a) {
  // (256 / per_batch + 2) runs x 3 NF columns must fit the 64 lanes of a wave
  return a.fold && a.slice_shift == 8 && !a.deterministic && a.tail.fin.gacc && a.P > 0 && a.P <= kMaxGradLDS &&
         (a.ev.order == 2 || a.ev.order == 4) && (256 / a.ev.per_batch + 2) * 3 * a.ev.order <= 64;
}
int launch_be_gather(const BeGatherArgs &a, int nb, hipStream_t s, hipEvent_t t0, hipEvent_t t1, hipEvent_t b0, hipEvent_t b1) {
  // (t0, t1) bracket the per-event kernel, (b0, b1) the per-batch pass that follows
  if (be_gather_folds(a)) {  // per-batch pass and finalize inside the four-events-per-lane kernel (see be_gather4_kernel)
    const dim3 g4(gather_blocks((a.ev.n + 3) / 4));
    if (a.ev.order == 2) CMX_LAUNCH(be_gather4_kernel<2>, g4, dim3(256), 0, s, t0, t1, a);
    else CMX_LAUNCH(be_gather4_kernel<4>, g4, dim3(256), 0, s, t0, t1, a);
    return 0;
  }
  if (a.slice_shift == 8) CMX_LAUNCH(be_gather4_kernel<0>, dim3(gather_blocks((a.ev.n + 3) / 4)), dim3(256), 0, s, t0, t1, a);
  else CMX_LAUNCH(be_gather_kernel, dim3(gather_blocks(a.ev.n)), dim3(256), 0, s, t0, t1, a);
  const int blocks = be_batch_blocks(nb);
  if (a.deterministic) {
    if (a.ev.order == 2) CMX_LAUNCH((be_gather_batch_kernel<2, true>), dim3(blocks), dim3(256), 0, s, b0, b1, a, nb);
    else CMX_LAUNCH((be_gather_batch_kernel<4, true>), dim3(blocks), dim3(256), 0, s, b0, b1, a, nb);
  } else {
    if (a.ev.order == 2) CMX_LAUNCH((be_gather_batch_kernel<2, false>), dim3(blocks), dim3(256), 0, s, b0, b1, a, nb);
    else CMX_LAUNCH((be_gather_batch_kernel<4, false>), dim3(blocks), dim3(256), 0, s, b0, b1, a, nb);
  }
  return blocks;
}

// ---------------------------------------------------------------------------------------------- K4 alpha
__global__ __launch_bounds__(256) void alpha_partials_kernel(AlphaArgs a) {
  __shared__ double red[4];
  double area_g = 0, num_g = 0, area_l = 0, num_l = 0, nz = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.npix; i += gridDim.x * 256) {
    const float gq = a.igp[i];
    const float il = a.il_old[i] + a.il_new[i];
    area_g += (double)(1.f - expf(-1.0f * gq));
    num_g += (double)gq;
    area_l += (double)(1.f - expf(-1.0f * il));
    num_l += (double)il;
    nz += (gq != 0.f) ? 1.0 : 0.0;
  }
  const double v[5] = {area_g, num_g, area_l, num_l, nz};
  for (int q = 0; q < 5; q++) {
    const double t = block_sum(v[q], red);
    if (threadIdx.x == 0) a.partials[(size_t)q * a.nblk + blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void alpha_finalize_kernel(AlphaArgs a) {
  __shared__ double red[4];
  double tot[5];
  for (int q = 0; q < 5; q++) {
    double s = 0;
    for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.partials[(size_t)q * a.nblk + i];
    tot[q] = block_sum(s, red);
  }
  if (threadIdx.x == 0) {
    double alpha = 0;
    if (tot[4] >= 1) alpha = (tot[3] / tot[2]) / (tot[1] / tot[0]);
    *a.alpha = alpha;
    *a.result_alpha = alpha;
  }
}
void launch_alpha(const AlphaArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(alpha_partials_kernel, dim3(a.nblk), dim3(256), 0, s, a);
  hipLaunchKernelGGL(alpha_finalize_kernel, dim3(1), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------- event store
// packed slot j = batch b (= j / per_batch), k-th sample of the batch: raw event b*B + k*rate  (event_pano_warper.cpp:188-196,262)
__global__ void be_pack_from_store_kernel(const uint32_t *raw, const long long *t, long long n, int B, int rate, int per_batch,
                                          int n_packed, long long t_next, uint32_t *out) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_packed; j += gridDim.x * blockDim.x) {
    const int b = j / per_batch, k = j - b * per_batch;
    const long long e = (long long)b * B + (long long)k * rate;
    out[j] = raw[e] | ((t[e] < t_next) ? 0x80000000u : 0u);
  }
}
void launch_be_pack_from_store(const uint32_t *raw, const long long *t, long long n, int B, int rate, int per_batch,
                               int n_packed, long long t_next, uint32_t *out, hipStream_t s) {
  hipLaunchKernelGGL(be_pack_from_store_kernel, dim3(2048), dim3(256), 0, s, raw, t, n, B, rate, per_batch, n_packed, t_next, out);
}

// Per-batch pose times of a window cut from the event store, on the device (the host loop over 50k batches costs more
// than a millisecond): time_batch = t_first + Duration((t_last - t_first).toSec() * 0.5) with ros::Duration's
// floor + round, exactly the host's time_batch_ns (event_pano_warper.cpp:239-242).  err[0] = first kind of error seen
// (CMX_ERR_TIME_ORDER / CMX_ERR_SPLINE_RANGE), err[1] = where.
__global__ void be_batch_times_kernel(const long long *t, long long n, int B, int nb, long long start_ns, long long dt_ns,
                                      int order, int K, long long *bt, long long *err) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const long long beg = (long long)b * B;
    const long long end = (n - beg > B) ? beg + B : n;
    const long long t_first = t[beg], t_last = t[end - 1];
    if (t_last < t_first) {
      if (atomicCAS((unsigned long long *)&err[0], 0ull, (unsigned long long)CMX_ERR_TIME_ORDER) == 0ull) err[1] = beg;
      continue;
    }
    const long long d = t_last - t_first;
    long long ds = d / 1000000000LL, dn = d % 1000000000LL;
    if (dn < 0) { dn += 1000000000LL; ds -= 1; }
    const double half = ((double)ds + 1e-9 * (double)dn) * 0.5;
    const long long hs = (long long)floor(half);
    const long long hn = (long long)round((half - (double)hs) * 1e9);
    const long long tb = t_first + hs * 1000000000LL + hn;
    const long long st = tb - start_ns;
    if (st < 0 || st / dt_ns + order > K) {
      if (atomicCAS((unsigned long long *)&err[0], 0ull, (unsigned long long)CMX_ERR_SPLINE_RANGE) == 0ull) err[1] = tb;
      continue;
    }
    bt[b] = tb;
  }
}
void launch_be_batch_times(const long long *t, long long n, int B, int nb, long long start_ns, long long dt_ns, int order, int K,
                           long long *bt, long long *err, hipStream_t s) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(be_batch_times_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, t, n, B, nb, start_ns, dt_ns, order, K, bt, err);
}

// ---------------------------------------------------------------------------------------------- Sobel moments
// contrast_ImageGradientMagnitude (reference local_focus_funcs.cpp:47-73): cv::Sobel 3x3 (REFLECT_101) of the
// blurred IWE and of each blurred derivative channel; contrast = mean(gx^2+gy^2), grad_k = 2 mean(gx*dgx_k + gy*dgy_k).
// Row filter then column filter in fp32, same order as the CPU path; fp64 sums.
__device__ __forceinline__ void sobel_at(const float *p, int W, int H, int x, int y, float &gx, float &gy) {
  const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W), ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
  const float *r0 = p + (size_t)ym * W, *r1 = p + (size_t)y * W, *r2 = p + (size_t)yp * W;
  // dx: row [-1 0 1], column [1 2 1]
  const float d0 = r0[xp] - r0[xm], d1 = r1[xp] - r1[xm], d2 = r2[xp] - r2[xm];
  gx = d0 + d1 * 2.f + d2;
  // dy: row [1 2 1], column [-1 0 1]
  const float s0 = r0[xm] + r0[x] * 2.f + r0[xp], s2 = r2[xm] + r2[x] * 2.f + r2[xp];
  gy = s2 - s0;
}

int sobel_blocks(int W, int H) {
  const int b = (W * H + 255) / 256;
  return b > 1024 ? 1024 : (b < 1 ? 1 : b);
}

__global__ __launch_bounds__(256) void sobel_moments_kernel(SobelArgs a) {
  __shared__ double red[4];
  const size_t np = (size_t)a.W * a.H;
  double acc0 = 0, acc[3] = {0, 0, 0};  // front end: P <= 3
  for (int i = blockIdx.x * 256 + threadIdx.x; i < (int)np; i += gridDim.x * 256) {
    const int y = i / a.W, x = i - y * a.W;
    float gx, gy;
    sobel_at(a.planes, a.W, a.H, x, y, gx, gy);
    const float hf = gx * gx + gy * gy;
    acc0 += (double)hf;
    for (int k = 0; k < a.P; k++) {
      float dgx, dgy;
      sobel_at(a.planes + (size_t)(1 + k) * np, a.W, a.H, x, y, dgx, dgy);
      const float m = gx * dgx + gy * dgy;
      acc[k] += (double)m;
    }
  }
  double t = block_sum(acc0, red);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
  for (int k = 0; k < a.P; k++) {
    t = block_sum(acc[k], red);
    if (threadIdx.x == 0) a.partials[(size_t)(1 + k) * a.nblk + blockIdx.x] = t;
  }
}
void launch_sobel_moments(const SobelArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(sobel_moments_kernel, dim3(a.nblk), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------- map upkeep
// EventWarper::updateIG (event_pano_warper.cpp:109-126): IG += IL_old where the visit count is <= max_update_times
__global__ void update_map_kernel(float *IG, const float *IL_old, const unsigned char *visits, int npix, int max_t) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x)
    if ((int)visits[i] <= max_t) IG[i] += IL_old[i];
}
void launch_update_map(float *IG, const float *IL_old, const unsigned char *visits, int npix, int max_update_times,
                       hipStream_t s) {
  hipLaunchKernelGGL(update_map_kernel, dim3(2048), dim3(256), 0, s, IG, IL_old, visits, npix, max_update_times);
}

// EventWarper::setUpdateTimesIG (event_pano_warper.cpp:81-107): warp every sensor pixel with one pose, mark the
// (2r+1)^2 neighbourhood of the hit cell (the reference's row test is `0 <= y_mask + j`; rows < 0, which the
// reference would write out of bounds, are skipped), then visits = saturate_u8(visits + mask).
struct RotArg { double R[9]; };
__global__ void mark_mask_kernel(BeSplatArgs a, RotArg rot, int sensor_h, int radius, unsigned char *mask) {
  const int npx = a.W * sensor_h;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const double *b = a.lut + 3 * (size_t)i;
    const double x = rot.R[0] * b[0] + rot.R[1] * b[1] + rot.R[2] * b[2];
    const double y = rot.R[3] * b[0] + rot.R[4] * b[1] + rot.R[5] * b[2];
    const double z = rot.R[6] * b[0] + rot.R[7] * b[1] + rot.R[8] * b[2];
    const double phi = atan2(x, z);
    const double theta = asin(y / sqrt(x * x + y * y + z * z));
    const int ic = (int)(a.cxp + phi * a.fx), ir = (int)(a.cyp + theta * a.fy);
    for (int di = -radius; di <= radius; di++)
      for (int dj = -radius; dj <= radius; dj++) {
        const int xm = ic + di, ym = ir + dj;
        if (0 <= ym + dj && ym < a.Hp && 0 <= xm && xm < a.Wp && ym >= 0) mask[(size_t)ym * a.Wp + xm] = 1;
      }
  }
}
__global__ void add_mask_kernel(unsigned char *visits, unsigned char *mask, int npix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    const int v = (int)visits[i] + (int)mask[i];
    visits[i] = (unsigned char)(v > 255 ? 255 : v);
    mask[i] = 0;
  }
}
void launch_mark_visited(const BeSplatArgs &cam, const double R[9], int sensor_h, int radius, unsigned char *mask,
                         unsigned char *visits, hipStream_t s) {
  RotArg rot;
  for (int i = 0; i < 9; i++) rot.R[i] = R[i];
  hipLaunchKernelGGL(mark_mask_kernel, dim3(1024), dim3(256), 0, s, cam, rot, sensor_h, radius, mask);
  hipLaunchKernelGGL(add_mask_kernel, dim3(2048), dim3(256), 0, s, visits, mask, cam.Wp * cam.Hp);
}

// planar [3][npix] -> interleaved [npix][3]  (CV_32FC3 layout of the reference's derivative image)
__global__ void interleave3_kernel(const float *planes, float *out, int npix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    out[3 * (size_t)i] = planes[i];
    out[3 * (size_t)i + 1] = planes[(size_t)npix + i];
    out[3 * (size_t)i + 2] = planes[2 * (size_t)npix + i];
  }
}
void launch_interleave3(const float *planes, float *out, int npix, hipStream_t s) {
  hipLaunchKernelGGL(interleave3_kernel, dim3(1024), dim3(256), 0, s, planes, out, npix);
}

}  // namespace cmx
