// cmx_binning.hip -- one-time (per packet / per window) sort of the events by DESTINATION tile, and the
// LDS-privatised splat kernels that consume the sorted order.
//
// Why: on gfx950 every device-scope fp32 atomic executes memory-side (the 8 XCD L2s are not coherent): one 32-byte
// fabric transaction per vote, ~18.7 G/s (profiles/r01a_*).  Motion-compensated events pile onto a few edge
// pixels, so a workgroup that owns a small image window can absorb thousands of votes per pixel in LDS
// (64-bit fixed-point ds_add, see below) and emit ONE global atomic per touched pixel.  Events are sorted by the 32x32 tile their vote lands
// in under the parameters of the first evaluation; during the solve the parameters move a little, so each
// workgroup's LDS window is the tile plus a 16-pixel margin, and any vote that still leaves the window takes the
// plain global-atomic path -- the result is exact for any parameters, only the speed depends on the binning.
// The sort is a counting sort (below); the (key, index) radix-sort kernels at the top of the file are its fallback for
// panoramas with more than 16 400 destination-tile keys.  The sorted order also carries per-event bearing / dt streams.
#include <mutex>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "cmx_internal.hpp"
#include "cmx_warp.hpp"
#include "cmx_tilepass.hpp"
#include "cmx_fusedgather.hpp"
#include "cmx_selfserve.hpp"

namespace cmx {

// Sort key = the 32x32 tile the vote lands in.  A vote that is NOT accepted under the binning parameters but lies within
// the window margin of the accepted region is keyed by the nearest accepted cell: when the parameters move it inside (a
// sensor-border event at omega = 0 a solve starts from; ~1 % of the events) it then finds itself in that tile's LDS
// window.  Left in the no-window sentinel these votes took the global-atomic path from ONE or two workgroups, which made
// them the kernel's critical path: 0.26 % of such votes cost the 1M-event splat 9.5 -> 15 us (profiles/r02_splat_drift.txt).
__device__ __forceinline__ uint32_t tile_key(int xx, int yy, bool ok, int W, int H, int tiles_x, uint32_t sentinel) {
  if (!ok) {
    if (xx < 1 - kBinMargin || xx >= W - 2 + kBinMargin || yy < 1 - kBinMargin || yy >= H - 2 + kBinMargin) return sentinel;
    xx = min(max(xx, 1), W - 3);
    yy = min(max(yy, 1), H - 3);
  }
  return (uint32_t)((yy / kBinTile) * tiles_x + xx / kBinTile);
}

__global__ __launch_bounds__(256) void fe_bin_keys_kernel(FeSplatArgs a, int tiles_x, int ntiles, uint32_t *keys,
                                                          uint32_t *idx) {
  fe_resolve_omega(a);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const FeWarp w = fe_warp_event<false>(a, i);
    keys[i] = tile_key(w.xx, w.yy, w.ok, a.W, a.H, tiles_x, (uint32_t)ntiles);
    idx[i] = (uint32_t)i;
  }
}
__global__ __launch_bounds__(256) void be_bin_keys_kernel(BeSplatArgs a, int tiles_x, int ntiles, uint32_t *keys,
                                                          uint32_t *idx) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const BeWarp w = be_warp_event<0>(a, i);
    // the IL_old / IL_new split is a property of the event (its timestamp), so it can be part of the sort key:
    // every chunk then votes into ONE plane and needs one LDS window
    const uint32_t t = tile_key(w.xx, w.yy, w.ok, a.Wp, a.Hp, tiles_x, (uint32_t)ntiles);
    keys[i] = t == (uint32_t)ntiles ? (uint32_t)(2 * ntiles) : 2 * t + (w.is_old ? 0u : 1u);
    idx[i] = (uint32_t)i;
  }
}
static int grid_for(int n) {
  int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}
void launch_fe_bin_keys(const FeSplatArgs &a, int tiles_x, int ntiles, uint32_t *keys, uint32_t *idx, hipStream_t s) {
  hipLaunchKernelGGL(fe_bin_keys_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a, tiles_x, ntiles, keys, idx);
}
void launch_be_bin_keys(const BeSplatArgs &a, int tiles_x, int ntiles, uint32_t *keys, uint32_t *idx, hipStream_t s) {
  hipLaunchKernelGGL(be_bin_keys_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a, tiles_x, ntiles, keys, idx);
}

int sort_pairs_u32(void *temp, size_t *temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                   uint32_t *vout, unsigned n, int end_bit, hipStream_t s) {
  return (int)rocprim::radix_sort_pairs(temp, *temp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}

__global__ __launch_bounds__(256) void apply_perm_kernel(const uint32_t *xy, const uint32_t *idx_sorted, int per_batch,
                                                         int n, uint32_t *sxy, uint32_t *sbatch) {
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const uint32_t i = idx_sorted[j];
    sxy[j] = xy[i];
    sbatch[j] = i / (uint32_t)per_batch;
  }
}
void launch_apply_perm(const uint32_t *xy, const uint32_t *idx_sorted, int per_batch, int n, uint32_t *sxy,
                       uint32_t *sbatch, hipStream_t s) {
  hipLaunchKernelGGL(apply_perm_kernel, dim3(grid_for(n)), dim3(256), 0, s, xy, idx_sorted, per_batch, n, sxy, sbatch);
}

// tile_start[t] = first sorted position whose key >= t   (t = 0 .. ntiles+1)
__global__ void tile_lower_bound_kernel(const uint32_t *keys, int n, int count, int *tile_start) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < (uint32_t)t) lo = mid + 1;
    else hi = mid;
  }
  tile_start[t] = lo;
}
void launch_tile_lower_bound(const uint32_t *keys_sorted, int n, int count, int *tile_start, hipStream_t s) {
  hipLaunchKernelGGL(tile_lower_bound_kernel, dim3((count + 255) / 256), dim3(256), 0, s, keys_sorted, n, count, tile_start);
}

// ---- counting sort by destination tile (the default: up to kCountSortMaxBins keys)
// The key space is tiny (300 tiles at 640x480, 2 x 1024 at 1024^2, 2 x 8192 at 4096x2048), so a comparison / radix
// sort of (key, index) pairs is the wrong tool: rocprim picks a merge sort here (21 launches, 167 us per 1M pairs).
// Four light passes instead, over contiguous slices of the (time-ordered) events, one slice per workgroup:
//   (1) keys + per-slice histogram: the warp kernel counts into an LDS histogram and stores it as one row of a
//       [slices][bins] table;
//   (2) one thread per bin turns its column into exclusive prefixes over the slices (and the bin's total);
//   (3) one workgroup scans the bin totals into tile_start;
//   (4) every workgroup loads tile_start[bin] + table[slice][bin] into LDS and scatters its slice, rank inside the
//       slice's run from an LDS atomic.
// A tile's events therefore stay in time order at slice granularity (4096 consecutive events), which is what keeps a
// wave's bearing-table and per-batch loads on a few cache lines: with runs reserved through global atomics, i.e. in
// arbitrary slice order, the splat ran 12.5-13.3 us instead of 11.3-11.5 us per 1M events.
constexpr int kCountSortMaxBins = 16400;
constexpr int kBinBlock = 1024;
constexpr int kBinMaxSlices = 512;
extern __shared__ int hist_sh[];

__device__ __forceinline__ void hist_zero(int nbins) {
  for (int k = threadIdx.x; k < nbins; k += kBinBlock) hist_sh[k] = 0;
  __syncthreads();
}
__device__ __forceinline__ void hist_store_row(int nbins, int *table) {
  __syncthreads();
  int *row = table + (size_t)blockIdx.x * nbins;
  for (int k = threadIdx.x; k < nbins; k += kBinBlock) row[k] = hist_sh[k];
}
__global__ __launch_bounds__(kBinBlock) void fe_bin_hist_kernel(FeSplatArgs a, int tiles_x, int ntiles, int per_block,
                                                                uint32_t *keys, int *table) {
  fe_resolve_omega(a);
  hist_zero(ntiles + 1);
  const int beg = blockIdx.x * per_block, end = min(a.n, beg + per_block);
  for (int i = beg + threadIdx.x; i < end; i += kBinBlock) {
    const FeWarp w = fe_warp_event<false>(a, i);
    const uint32_t key = tile_key(w.xx, w.yy, w.ok, a.W, a.H, tiles_x, (uint32_t)ntiles);
    keys[i] = key;
    atomicAdd(&hist_sh[key], 1);
  }
  hist_store_row(ntiles + 1, table);
}
__global__ __launch_bounds__(kBinBlock) void be_bin_hist_kernel(BeSplatArgs a, int tiles_x, int ntiles, int per_block,
                                                                uint32_t *keys, int *table) {
  hist_zero(2 * ntiles + 1);
  const int beg = blockIdx.x * per_block, end = min(a.n, beg + per_block);
  for (int i = beg + threadIdx.x; i < end; i += kBinBlock) {
    const BeWarp w = be_warp_event<0>(a, i);
    const uint32_t t = tile_key(w.xx, w.yy, w.ok, a.Wp, a.Hp, tiles_x, (uint32_t)ntiles);
    const uint32_t key = t == (uint32_t)ntiles ? (uint32_t)(2 * ntiles) : 2 * t + (w.is_old ? 0u : 1u);
    keys[i] = key;
    atomicAdd(&hist_sh[key], 1);
  }
  hist_store_row(2 * ntiles + 1, table);
}
// one thread per bin: table[s][k] <- number of events with key k in slices < s; total[k] = the bin's count
__global__ __launch_bounds__(256) void column_scan_kernel(int *table, int nbins, int nslices, int *total) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nbins) return;
  int run = 0, s = 0;
  for (; s + 8 <= nslices; s += 8) {  // the loads do not depend on the running sum: eight in flight
    int v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = table[(size_t)(s + q) * nbins + k];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      table[(size_t)(s + q) * nbins + k] = run;
      run += v[q];
    }
  }
  for (; s < nslices; s++) {
    const int v = table[(size_t)s * nbins + k];
    table[(size_t)s * nbins + k] = run;
    run += v;
  }
  total[k] = run;
}
// one workgroup: tile_start[t] = number of events with key < t for t = 0 .. nbins (nbins + 1 entries)
__global__ __launch_bounds__(kBinBlock) void scan_bins_kernel(const int *total, int nbins, int *tile_start) {
  __shared__ int wave_tot[kBinBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nbins + kBinBlock - 1) / kBinBlock;
  const int k0 = min(nbins, tid * per), k1 = min(nbins, k0 + per);
  int sum = 0;
  for (int k = k0; k < k1; k++) sum += total[k];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int off = incl - sum;
  for (int w = 0; w < wave; w++) off += wave_tot[w];
  for (int k = k0; k < k1; k++) {
    tile_start[k] = off;
    off += total[k];
  }
  if (tid == kBinBlock - 1) tile_start[nbins] = off;  // the last thread's running offset is the total (empty segments add 0)
}
// sb / sdt (front end, optional): the bearing and the batch dt of every event are gathered HERE, once per packet, and
// stored in sorted order: the per-evaluation kernels then stream them (coalesced) instead of gathering from the
// bearing table and the dt table with 64 different addresses per wave -- on gfx950 such a divergent load costs a
// kernel ~3 us per 1M events whatever the cache hit rate (the texture-address path handles one lane's line per clock)
__global__ __launch_bounds__(kBinBlock) void scatter_bins_kernel(const uint32_t *keys, const uint32_t *xy, int per_batch, int n,
                                                                 int nbins, int per_block, const int *table,
                                                                 const int *tile_start, uint32_t *sxy, uint32_t *sbatch,
                                                                 const double *lut2, int W, const double *batch_dt, double *sb,
                                                                 double *sdt) {
  const int *row = table + (size_t)blockIdx.x * nbins;
  for (int k = threadIdx.x; k < nbins; k += kBinBlock) hist_sh[k] = tile_start[k] + row[k];  // where this slice's run starts
  __syncthreads();
  const int beg = blockIdx.x * per_block, end = min(n, beg + per_block);
  for (int i = beg + threadIdx.x; i < end; i += kBinBlock) {
    const int pos = atomicAdd(&hist_sh[keys[i]], 1);
    const uint32_t e = xy[i], bi = (uint32_t)i / (uint32_t)per_batch;
    sxy[pos] = e;
    sbatch[pos] = bi;
    if (sb) {
      const double2 v = *reinterpret_cast<const double2 *>(lut2 + 2 * ((size_t)((e >> 16) & 0x7fff) * W + (e & 0xffff)));
      *reinterpret_cast<double2 *>(sb + 2 * (size_t)pos) = v;
      if (sdt) sdt[pos] = batch_dt[bi];
    }
  }
}
bool count_sort_ok(int nbins) { return nbins <= kCountSortMaxBins; }
static void bin_grid(int n, int &blocks, int &per_block) {
  blocks = (n + 4095) / 4096;
  blocks = blocks < 1 ? 1 : (blocks > kBinMaxSlices ? kBinMaxSlices : blocks);
  per_block = ((n + blocks - 1) / blocks + kBinBlock - 1) / kBinBlock * kBinBlock;
  blocks = (n + per_block - 1) / per_block;  // slices that actually hold events
}
size_t count_sort_scratch_ints(int n, int nbins) {  // [bin totals | slices x bins table]
  int blocks, per_block;
  bin_grid(n, blocks, per_block);
  return (size_t)nbins * (size_t)(blocks + 1);
}
// (per DEVICE: a group's members sit on several devices of one process and launch from concurrent worker threads -- the attribute
//  is set once on each device, behind that device's own once-flag)
static void allow_big_lds() {  // 16400 bins x 4 B is just above the 64 KB default of dynamic LDS
  constexpr int kMaxDev = 64;
  static std::once_flag done[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  auto set = [] {
    const int bytes = kCountSortMaxBins * (int)sizeof(int);
    hipFuncSetAttribute((const void *)fe_bin_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipFuncSetAttribute((const void *)be_bin_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipFuncSetAttribute((const void *)scatter_bins_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  };
  if (dev < kMaxDev) std::call_once(done[dev], set);
  else set();
}
// keys: n u32 scratch; scratch: count_sort_scratch_ints(n, nbins) ints (contents irrelevant); tile_start: nbins + 1 ints
void launch_count_sort(const FeSplatArgs *fe, const BeSplatArgs *be, int tiles_x, int ntiles_img, const uint32_t *xy,
                       int per_batch, int n, uint32_t *keys, int *scratch, int *tile_start, uint32_t *sxy, uint32_t *sbatch,
                       double *sb, double *sdt, hipStream_t s) {
  allow_big_lds();
  const int nbins = (fe ? ntiles_img : 2 * ntiles_img) + 1;
  int blocks, per_block;
  bin_grid(n, blocks, per_block);
  int *total = scratch, *table = scratch + nbins;
  const size_t lds = (size_t)nbins * sizeof(int);
  if (fe) hipLaunchKernelGGL(fe_bin_hist_kernel, dim3(blocks), dim3(kBinBlock), lds, s, *fe, tiles_x, ntiles_img, per_block, keys, table);
  else hipLaunchKernelGGL(be_bin_hist_kernel, dim3(blocks), dim3(kBinBlock), lds, s, *be, tiles_x, ntiles_img, per_block, keys, table);
  hipLaunchKernelGGL(column_scan_kernel, dim3((nbins + 255) / 256), dim3(256), 0, s, table, nbins, blocks, total);
  hipLaunchKernelGGL(scan_bins_kernel, dim3(1), dim3(kBinBlock), 0, s, total, nbins, tile_start);
  const double *lut2 = fe ? fe->lut2 : be->lut2;
  const bool streams = sb && lut2 && (be || sdt);
  hipLaunchKernelGGL(scatter_bins_kernel, dim3(blocks), dim3(kBinBlock), lds, s, keys, xy, per_batch, n, nbins, per_block, table,
                     tile_start, sxy, sbatch, streams ? lut2 : nullptr, fe ? fe->W : be->W, fe ? fe->batch_dt : nullptr,
                     streams ? sb : nullptr, (streams && fe) ? sdt : nullptr);
}

// Chunk table on the device: tile t owns the sorted events [tile_start[t], tile_start[t+1]); it is cut into chunks of at
// most M events.  One workgroup.  Order of the table = order the workgroups start in = longest-processing-time first:
// all full chunks, then the remainder chunks by falling size (64 size classes) -- with an unsorted tail the back-end splat
// ran 68 us instead of 52.  kRankSortMax: tile offsets up to that many are staged in LDS.  Entry
// `ntiles` is the sentinel tile of events whose vote is not accepted under the binning parameters: no LDS window.
constexpr int kRankSortMax = 4096;
constexpr int kSentinelChunk = 256;  // == cmx_internal.hpp's bound in do_binning (max_chunks)
__global__ __launch_bounds__(1024) void build_chunks_kernel(const int *tile_start_g, int ntiles, int planes_per_tile, int tiles_x,
                                                            int margin, int M, Chunk *chunks, int *count,
                                                            unsigned long long *count_host, unsigned binning_id, FusedTables fused) {
  __shared__ int wave_tot[16];
  __shared__ int base_sh;
  __shared__ int ts_sh[kRankSortMax + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = ntiles + 1;  // tiles including the sentinel
  // one workgroup, many dependent reads of the offsets: keep them in LDS when they fit (a global read here is a ~1 us
  // round trip to another XCD's L2 with nothing to overlap it)
  const int *tile_start = tile_start_g;
  if (T + 1 <= kRankSortMax + 2) {
    for (int t = tid; t <= T; t += 1024) ts_sh[t] = tile_start_g[t];
    __syncthreads();
    tile_start = ts_sh;
  }
  // the no-window sentinel's events take the global-atomic path if they become valid: small chunks, so that this serial
  // work is spread over many workgroups instead of forming the tail of the launch
  auto Mof = [&](int t) { return t == ntiles ? min(M, kSentinelChunk) : M; };
  auto make_chunk = [&](int t, int beg, int end) {
    const bool sentinel = (t == ntiles);
    const int tile = t / planes_per_tile, plane = t % planes_per_tile;
    const int wx0 = sentinel ? -200000000 : (tile % tiles_x) * kBinTile - margin;
    const int wy0 = sentinel ? -200000000 : (tile / tiles_x) * kBinTile - margin;
    return Chunk{wx0, wy0, beg, end, plane, sentinel ? -1 : tile};
  };
  if (tid == 0) base_sh = 0;
  __syncthreads();
  // full chunks of every tile, in tile order (block-wide exclusive scan of the per-tile counts)
  for (int t0 = 0; t0 < T; t0 += 1024) {
    const int t = t0 + tid;
    int beg = 0, nfull = 0;
    if (t < T) {
      beg = tile_start[t];
      int len = tile_start[t + 1] - beg;
      if (len < 0) len = 0;
      nfull = len / Mof(t);
    }
    int incl = nfull;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = base_sh + incl - nfull, tot = 0;
    for (int w = 0; w < 16; w++) {
      if (w < wave) off += wave_tot[w];
      tot += wave_tot[w];
    }
    for (int k = 0; k < nfull; k++) chunks[off + k] = make_chunk(t, beg + k * Mof(t), beg + (k + 1) * Mof(t));
    __syncthreads();
    if (tid == 0) base_sh += tot;
    __syncthreads();
  }
  const int nfull_total = base_sh;
  // Remainder chunks (one per tile that has one) in FALLING SIZE order, so that the big ones start first: 64 size classes
  // (class 0 = the largest sixty-fourth of a chunk), order inside a class arbitrary.  Two passes over the tiles -- count per
  // class, exclusive scan of the 64 counts, place -- instead of an exact sort: the bitonic sort in LDS this replaces was 78
  // barrier-separated stages (75 us of a window's first evaluation at 2049 tiles), the five-class form before it cost the
  // back-end splat 3 us of load balance.  (Which chunk of a class comes first varies from run to run; every sum the splat
  // makes is order-independent in CMX_OPT_DETERMINISTIC.)
  __shared__ int ccount[64], cbase[64], cplace[64];
  if (tid < 64) { ccount[tid] = 0; cplace[tid] = 0; }
  __syncthreads();
  auto class_of = [&](int rem, int m) { const int b = (int)(((long long)rem * 64) / m); return 63 - (b > 63 ? 63 : b); };
  for (int t = tid; t < T; t += 1024) {
    int len = tile_start[t + 1] - tile_start[t];
    if (len < 0) len = 0;
    const int rem = len % Mof(t);
    if (rem) atomicAdd(&ccount[class_of(rem, Mof(t))], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < 64; k++) { cbase[k] = run; run += ccount[k]; }
    *count = nfull_total + run;
    // (binning id, count) as ONE 8-byte store: the host accepts the count only with the id of the binning it is waiting for
    if (count_host) *count_host = ((unsigned long long)binning_id << 32) | (unsigned)(nfull_total + run);
  }
  __syncthreads();
  for (int t = tid; t < T; t += 1024) {
    const int beg = tile_start[t];
    int len = tile_start[t + 1] - beg;
    if (len < 0) len = 0;
    const int rem = len % Mof(t);
    if (rem) {
      const int k = class_of(rem, Mof(t));
      chunks[nfull_total + cbase[k] + atomicAdd(&cplace[k], 1)] = make_chunk(t, beg + (len / Mof(t)) * Mof(t), beg + len);
    }
  }
  // tables of the fused splat + image pass (FusedArgs): how many chunk arrivals complete each tile's 5 x 5 neighbourhood; the
  // arrival counters and the moment rows start from zero (rows of tiles nobody runs stay zero for as long as this table lives)
  if (fused.nbr_expected) {  // (front end: one plane per tile; the launcher checks ntiles <= kRankSortMax)
    __shared__ int nact_sh;
    if (tid == 0) nact_sh = 0;
    int *nch = ts_sh;  // the offsets are not needed any more once every thread has formed its own counts
    int mine[(kRankSortMax + 1023) / 1024];
#pragma unroll
    for (int q = 0; q < (kRankSortMax + 1023) / 1024; q++) {
      const int t = tid + q * 1024;
      int len = t < ntiles ? tile_start[t + 1] - tile_start[t] : 0;
      if (len < 0) len = 0;
      mine[q] = len / M + (len % M ? 1 : 0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (kRankSortMax + 1023) / 1024; q++)
      if (tid + q * 1024 < ntiles) nch[tid + q * 1024] = mine[q];
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024) {
      const int tx = t % tiles_x, ty = t / tiles_x;
      int sum = 0;
      for (int dy = -kFuseNbr; dy <= kFuseNbr; dy++)
        for (int dx = -kFuseNbr; dx <= kFuseNbr; dx++) {
          const int ux = tx + dx, uy = ty + dy;
          if (ux >= 0 && ux < tiles_x && uy >= 0 && uy < fused.tiles_y) sum += nch[uy * tiles_x + ux];
        }
      for (int h = 0; h < kFuseStrips; h++) {
        const int u = t * kFuseStrips + h;
        fused.nbr_expected[u] = sum;
        fused.nbr_cnt[(size_t)u * kFuseCntStride] = 0u;
        fused.partials[u] = 0.0;
        fused.partials[ntiles * kFuseStrips + u] = 0.0;
      }
      if (sum > 0) atomicAdd(&nact_sh, kFuseStrips);
    }
    __syncthreads();
    if (tid == 0) {
      if (fused.n_active) *fused.n_active = nact_sh;
      if (fused.tiles_done) *fused.tiles_done = 0u;
    }
  }
}
bool fused_tables_ok(int ntiles, int planes_per_tile) { return planes_per_tile == 1 && ntiles + 2 <= kRankSortMax + 2; }
void launch_build_chunks(const int *tile_start, int ntiles, int planes_per_tile, int tiles_x, int margin, int M, Chunk *chunks,
                         int *count, unsigned long long *count_host, unsigned binning_id, hipStream_t s, const FusedTables *fused) {
  FusedTables ft{};
  if (fused && fused_tables_ok(ntiles, planes_per_tile)) ft = *fused;
  hipLaunchKernelGGL(build_chunks_kernel, dim3(1), dim3(1024), 0, s, tile_start, ntiles, planes_per_tile, tiles_x, margin, M,
                     chunks, count, count_host, binning_id, ft);
}

// ---------------------------------------------------------------------------------------------- LDS splats
// One workgroup = one chunk of sorted events.  win: kBinWindow^2 fp32 per plane.
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
__global__ __launch_bounds__(256) void fixed_to_float_kernel(fix_t *fixed, float *planes, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const fix_t v = fixed[i];
    if (v != 0ull) {
      planes[i] = (float)((double)v * kFixInv);
      fixed[i] = 0ull;
    }
  }
}
void launch_fixed_to_float(unsigned long long *fixed, float *planes, size_t n, hipStream_t s) {
  if (n == 0) return;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fixed, planes, n);
}

static_assert(kBinWindow * kBinWindow % 256 == 0, "window cells per thread");
constexpr int kUnroll = 2;  // events in flight per thread (swept on MI355X: 2 -> 12.6 us, 1 -> 13.3, 4 -> 14.1, 8 -> 15.2 per 1M events)

#ifndef CMX_FE_SPLAT_NT
#define CMX_FE_SPLAT_NT 512  // threads per chunk workgroup (round 5 sweep, profiles/r05_fe_shape.txt: 512 threads x chunks of n / 256
                            // events: splat 8.7 -> 7.9 us, a 1M-event solve 0.556 -> 0.535 ms, a 60k-event solve 0.84 -> 0.80 ms)
#endif
constexpr int kFeSplatNT = CMX_FE_SPLAT_NT;
static_assert(kBinWindow * kBinWindow % kFeSplatNT == 0, "window cells per thread");
static_assert(sizeof(fix_t) * kBinWindow * kBinStride >= kTpLdsBytes, "the fused tile pass reuses the vote window's LDS");
constexpr unsigned long long kFuseTimeoutTicks = 200000ull;  // 2 ms of the 100 MHz wall clock: ~200 x the launch's own duration
// FUSE 1: the adjoint image pass runs inside this launch, tile by tile, as the tiles' inputs complete (FusedArgs, cmx_tilepass.hpp)
// FUSE 2: ... and so do the gradient gather and the finalize step (cmx_fusedgather.hpp): one launch per evaluation
// FUSE 3: one launch of the chunk workgroups alone: each runs the passes of the tiles it owns and gathers its own events (cmx_selfserve.hpp)
template <bool FIXED, bool STREAM, int FUSE>
__device__ __forceinline__ void fe_splat_lds_body(FeSplatArgs &a, const BinnedEvents &b, const FusedArgs &f) {
  __shared__ __attribute__((aligned(16))) fix_t win[kBinWindow * kBinStride];
  if (wg_stop_requested(a.skip)) return;  // device-driven solve: finished
  if (FUSE == 2 && (int)blockIdx.x >= b.nchunks + f.tiles_x * f.tiles_y * kFuseStrips) {  // GATHER ROLE
    __shared__ FgSmem fg_sm;
    fused_gather_role<kFeSplatNT>(a, b, f, (int)blockIdx.x - b.nchunks - f.tiles_x * f.tiles_y * kFuseStrips, fg_sm);
    return;
  }
  if ((FUSE == 1 || FUSE == 2) && (int)blockIdx.x >= b.nchunks) {
    // TILE ROLE: the workgroups behind the chunk table's launch bound each own one 32 x 32 image tile.  They are dispatched after
    // every chunk workgroup (lower indices), wait -- one polling lane, the other waves parked at the barrier -- until the chunks
    // that can vote into the tile's neighbourhood have all arrived, and run the tile's image pass.  A wait only ever points at
    // workgroups dispatched EARLIER that wait for nobody; it is bounded all the same (kFuseTimeoutTicks): a tile that gives up
    // reports through the fallback counter and the host repeats the evaluation through the separate launches.
    const int t = (int)blockIdx.x - b.nchunks;
    const unsigned expected = (unsigned)f.nbr_expected[t];
    if (f.trace && threadIdx.x == 0) { f.trace[8 * (size_t)blockIdx.x] = wall_clock64(); f.trace[8 * (size_t)blockIdx.x + 3] = expected ? 2 : 3; }
    if (expected == 0u || (f.debug & 1)) return;  // no vote can reach this tile: B = Jt = 0 there, zero moments (rows cleared at sort time)
    __shared__ int ok_sh;
    auto wait_inputs = [&]() -> bool {
      if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(f.nbr_cnt + (size_t)t * kFuseCntStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
          if (f.debug & 8) __builtin_amdgcn_s_sleep(32); else if (f.debug & 16) __builtin_amdgcn_s_sleep(127); else __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > kFuseTimeoutTicks) { ok = 0; break; }
        }
        if (ok) __hip_atomic_store(f.nbr_cnt + (size_t)t * kFuseCntStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all-zero again for the next launch
        else atomicOr(b.fallback, kFuseIncomplete);
        if (f.trace) f.trace[8 * (size_t)blockIdx.x + 1] = wall_clock64();
        ok_sh = ok;
      }
      __syncthreads();
      return ok_sh != 0 && !(f.debug & 2);
    };
    fused_tile_pass<kFeSplatNT, FUSE == 2>(f, a.planes, a.W, a.H, t, reinterpret_cast<unsigned char *>(win), wait_inputs);
    if (FUSE == 2) {  // publish: every wave's write-through stores have left, then the strip's stamp and the launch's strip count
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(f.tile_done + (size_t)t * kFuseCntStride, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(f.tiles_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (f.trace && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); f.trace[8 * (size_t)blockIdx.x + 2] = wall_clock64(); }
    return;
  }
  fe_resolve_omega(a);
  // the launch is sized by an upper bound of the table's length, and so is the table's allocation: the entry is read
  // BEFORE the length is checked, so that the two loads share one memory round trip instead of taking two (~1 us each)
  Chunk c = b.chunks[blockIdx.x];
  if (FUSE && f.trace && threadIdx.x == 0) { f.trace[8 * (size_t)blockIdx.x] = wall_clock64(); f.trace[8 * (size_t)blockIdx.x + 3] = 0; }
  const bool has_chunk = (int)blockIdx.x < *b.nchunks_dev;
  if (!has_chunk) {
    if (FUSE != 3) return;
    c = Chunk{-200000000, -200000000, 0, 0, 0, -1};  // (self-service: a workgroup beyond the table still owns tiles and arrives)
  }
  const bool has_win = c.wx0 > -100000000;
  const int tid = threadIdx.x;
  // votes on the global path are counted per workgroup (LDS) and reported with ONE device atomic: a counter every thread
  // adds to is a single memory-side address -- ~1.3 ns per add, 13 us per percent of a million events' votes
  __shared__ unsigned sfall;
  if (tid == 0) sfall = 0;
  if (has_win)
    for (int p = tid; p < kBinWindow * kBinStride; p += kFeSplatNT) win[p] = 0ull;
  __syncthreads();
  unsigned nfall = 0;
  for (int j0 = c.beg + tid; j0 < c.end; j0 += kFeSplatNT * kUnroll) {
    bool act[kUnroll];
    double px[kUnroll], py[kUnroll], pz[kUnroll], dt[kUnroll];
    if (STREAM) {  // bearing and dt of every sorted event stream in (coalesced): no table gathers in this kernel
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const int j = j0 + u * kFeSplatNT;
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
        const int j = j0 + u * kFeSplatNT;
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
          // beyond the reach the tiles' arrival counts cover (kFuseReach px around the chunk's tile; window origin = tile - margin)
          if (!has_win || lx < kBinMargin - kFuseReach || lx + 1 > kBinMargin + kBinTile - 1 + kFuseReach || ly < kBinMargin - kFuseReach ||
              ly + 1 > kBinMargin + kBinTile - 1 + kFuseReach)
            nfall |= kFuseUnsafe;
        }
      }
    }
  }
  if (nfall) {
    atomicAdd(&sfall, nfall & kFuseCountMask);
    if (nfall & kFuseUnsafe) atomicOr(&sfall, kFuseUnsafe);
  }
  __syncthreads();
  if (tid == 0 && sfall) {
    if (sfall & kFuseCountMask) atomicAdd(b.fallback, sfall & kFuseCountMask);
    if (sfall & kFuseUnsafe) atomicOr(b.fallback, kFuseUnsafe);
  }
  if (has_win) {
    // all of a thread's window cells are read before the first is flushed: one LDS round trip instead of sixteen
    // (the rolled loop waited for every read in turn: ~0.9 of the kernel's ~9 us, profiles/r02_splat_timeline.txt)
    constexpr int kCells = kBinWindow * kBinWindow / kFeSplatNT;
    fix_t cell[kCells];
#pragma unroll
    for (int k = 0; k < kCells; k++) {
      const int p = tid + kFeSplatNT * k, ly = p / kBinWindow, lx = p - ly * kBinWindow;
      cell[k] = win[ly * kBinStride + lx];
    }
#pragma unroll
    for (int k = 0; k < kCells; k++) {
      const fix_t v = cell[k];
      if (v != 0ull) {
        const int p = tid + kFeSplatNT * k, ly = p / kBinWindow, lx = p - ly * kBinWindow;
        const size_t at = (size_t)(c.wy0 + ly) * a.W + (c.wx0 + lx);
        if (FIXED) atomicAdd(b.fixed + at, v);
        else atomic_add_f32(a.planes + at, (float)((double)v * kFixInv));
      }
    }
  }
  if (FUSE) {
    // This chunk's votes are on their way to the plane as agent-scope atomics: drain them, then arrive on the strips of the (up to)
    // 25 tiles whose image pass reads pixels this chunk can have touched -- one lane, one fire-and-forget atomic each.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's atomics have been performed
    constexpr int kSide = 2 * kFuseNbr + 1;
    if (c.tile >= 0 && tid < kSide * kSide * kFuseStrips && !(f.debug & 4)) {
      const int q = tid / kFuseStrips, h = tid % kFuseStrips;
      const int tx = c.tile % f.tiles_x + (q % kSide - kFuseNbr), ty = c.tile / f.tiles_x + (q / kSide - kFuseNbr);
      if (tx >= 0 && tx < f.tiles_x && ty >= 0 && ty < f.tiles_y)
        __hip_atomic_fetch_add(f.nbr_cnt + (size_t)((ty * f.tiles_x + tx) * kFuseStrips + h) * kFuseCntStride, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (f.trace && tid == 0) { f.trace[8 * (size_t)blockIdx.x + 1] = wall_clock64(); f.trace[8 * (size_t)blockIdx.x + 3] = 1; }
  }
  if constexpr (FUSE == 3 && STREAM) {
    __shared__ SsSmem ss_sm;
    // (behind the barrier above: the window's cells are in registers or flushed, sfall is final)
    self_serve_tail<kFeSplatNT>(a, b, f, c, has_chunk, (sfall & kFuseCountMask) != 0u, reinterpret_cast<unsigned char *>(win), ss_sm);
  }
}
template <bool FIXED, bool STREAM, int FUSE>
__global__ __launch_bounds__(kFeSplatNT) void fe_splat_lds_kernel(FeSplatArgs a, BinnedEvents b, FusedArgs f) {
  fe_splat_lds_body<FIXED, STREAM, FUSE>(a, b, f);
}
// the one-launch form (FUSE = 2): at least six waves per SIMD = three 512-thread workgroups per CU -- the chunk and strip workgroups of
// a 1M-event launch (413 + 300) must all be resident at once, and the gather role's registers would otherwise take the kernel to 93
// VGPRs = two workgroups per CU
template <bool STREAM>
__global__ __launch_bounds__(kFeSplatNT) __attribute__((amdgpu_waves_per_eu(6, 8))) void fe_splat_lds_one_kernel(FeSplatArgs a, BinnedEvents b,
                                                                                                                FusedArgs f) {
  fe_splat_lds_body<false, STREAM, 2>(a, b, f);
}
// the self-service one-launch form (FUSE = 3): every workgroup of the launch must be resident at once -- at least four waves per SIMD
// = two 512-thread workgroups per CU (<= 128 VGPRs): 512 chunk workgroups on 256 CUs (a 1M-event packet has ~413)
__global__ __launch_bounds__(kFeSplatNT) __attribute__((amdgpu_waves_per_eu(4, 8))) void fe_splat_lds_self_kernel(FeSplatArgs a, BinnedEvents b,
                                                                                                                 FusedArgs f) {
  fe_splat_lds_body<false, true, 3>(a, b, f);
}
int fe_selfserve_capacity() {
  static std::mutex mu;
  static int cap[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lk(mu);
  if (cap[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fe_splat_lds_self_kernel, kFeSplatNT, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cap[dev] = -1;
    else cap[dev] = per_cu * cus > 0 ? per_cu * cus : -1;
  }
  return cap[dev] > 0 ? cap[dev] : 0;
}
template <bool FIXED, bool STREAM, int FUSE>
static void launch_fe_splat_lds_t(const FeSplatArgs &a, const BinnedEvents &b, const FusedArgs &f, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  // chunk workgroups, then one workgroup per image strip, then (one-launch evaluation) the gather workgroups
  const dim3 grid(b.nchunks + (FUSE ? f.tiles_x * f.tiles_y * kFuseStrips : 0) + (FUSE == 2 ? f.gather_blocks : 0));
  if constexpr (FUSE == 2) {
    if (t0 || t1) hipExtLaunchKernelGGL((fe_splat_lds_one_kernel<STREAM>), grid, dim3(kFeSplatNT), 0, s, t0, t1, 0, a, b, f);
    else hipLaunchKernelGGL((fe_splat_lds_one_kernel<STREAM>), grid, dim3(kFeSplatNT), 0, s, a, b, f);
  } else {
    if (t0 || t1) hipExtLaunchKernelGGL((fe_splat_lds_kernel<FIXED, STREAM, FUSE>), grid, dim3(kFeSplatNT), 0, s, t0, t1, 0, a, b, f);
    else hipLaunchKernelGGL((fe_splat_lds_kernel<FIXED, STREAM, FUSE>), grid, dim3(kFeSplatNT), 0, s, a, b, f);
  }
}
void launch_fe_splat_lds(const FeSplatArgs &a, const BinnedEvents &b, hipStream_t s, hipEvent_t t0, hipEvent_t t1, const FusedArgs *fused) {
  if (b.nchunks <= 0) return;
  const bool stream = b.sb && b.sdt;
  const FusedArgs none{};
  if (fused && !b.fixed && fused->self_serve && stream) {  // one launch of the chunk workgroups alone (cmx_selfserve.hpp)
    const dim3 grid(b.nchunks);
    if (t0 || t1) hipExtLaunchKernelGGL(fe_splat_lds_self_kernel, grid, dim3(kFeSplatNT), 0, s, t0, t1, 0, a, b, *fused);
    else hipLaunchKernelGGL(fe_splat_lds_self_kernel, grid, dim3(kFeSplatNT), 0, s, a, b, *fused);
  } else if (fused && !b.fixed && fused->gather_blocks > 0 && stream) {  // one launch per evaluation (needs the tile-ordered streams)
    launch_fe_splat_lds_t<false, true, 2>(a, b, *fused, s, t0, t1);
  } else if (fused && !b.fixed) {  // (never with the deterministic mode's fixed-point planes)
    if (stream) launch_fe_splat_lds_t<false, true, 1>(a, b, *fused, s, t0, t1);
    else launch_fe_splat_lds_t<false, false, 1>(a, b, *fused, s, t0, t1);
  } else if (b.fixed) {
    if (stream) launch_fe_splat_lds_t<true, true, 0>(a, b, none, s, t0, t1);
    else launch_fe_splat_lds_t<true, false, 0>(a, b, none, s, t0, t1);
  } else {
    if (stream) launch_fe_splat_lds_t<false, true, 0>(a, b, none, s, t0, t1);
    else launch_fe_splat_lds_t<false, false, 0>(a, b, none, s, t0, t1);
  }
}

template <bool FIXED>
__global__ __launch_bounds__(256) void be_splat_lds_kernel(BeSplatArgs a, BinnedEvents b) {
  __shared__ fix_t win[kBinWindow * kBinStride];  // one plane per chunk: the sort key separates IL_old / IL_new events
  // the launch is sized by an upper bound of the table's length, and so is the table's allocation: the entry is read
  // BEFORE the length is checked, so that the two loads share one memory round trip instead of taking two (~1 us each)
  const Chunk c = b.chunks[blockIdx.x];
  if ((int)blockIdx.x >= *b.nchunks_dev) return;
  const bool has_win = c.wx0 > -100000000;
  const int tid = threadIdx.x;
  const size_t np = (size_t)a.Wp * a.Hp;
  // votes on the global path are counted per workgroup (LDS) and reported with ONE device atomic: a counter every thread
  // adds to is a single memory-side address -- ~1.3 ns per add, 13 us per percent of a million events' votes
  __shared__ unsigned sfall;
  if (tid == 0) sfall = 0;
  if (has_win)
    for (int p = tid; p < kBinWindow * kBinStride; p += 256) win[p] = 0ull;
  __syncthreads();
  unsigned nfall = 0;
  constexpr int U = 2;
  for (int j0 = c.beg + tid; j0 < c.end; j0 += 256 * U) {
    uint32_t e[U], bi[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = j0 + u * 256;
      act[u] = j < c.end;
      e[u] = act[u] ? b.sxy[j] : 0u;
      bi[u] = act[u] ? b.sbatch[j] : 0u;
    }
    double rx[U], ry[U], rz[U];  // e_ray_w = R * bearing of the U events in flight
#pragma unroll
    for (int u = 0; u < U; u++) {
      double R[9];
      const double *Rp = a.poseR[bi[u]].R;
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rp[k];
      if (b.sb) {  // tile-ordered bearing stream: (x, y), z = 1
        const int jj = act[u] ? j0 + u * 256 : c.beg;
        const double2 v = *reinterpret_cast<const double2 *>(b.sb + 2 * (size_t)jj);
        be_rotate<true>(R, v.x, v.y, 1.0, rx[u], ry[u], rz[u]);
      } else {
        double b0, b1, b2;
        load_bearing(a, (int)(e[u] & 0xffff), (int)((e[u] >> 16) & 0x7fff), b0, b1, b2);
        be_rotate<false>(R, b0, b1, b2, rx[u], ry[u], rz[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const BeWarp w = be_project<0>(a, e[u], (int)bi[u], rx[u], ry[u], rz[u]);
      if (act[u] && w.ok) {
        const int lx = w.xx - c.wx0, ly = w.yy - c.wy0;
        if (has_win && lx >= 0 && lx < kBinWindow - 1 && ly >= 0 && ly < kBinWindow - 1) {
          vote4_lds(win, lx, ly, w.dx, w.dy);
        } else {
          if (FIXED) vote4_global_fix(b.fixed + (w.is_old ? 0 : np), a.Wp, w.xx, w.yy, w.dx, w.dy);
          else vote4_global(a.planes + (w.is_old ? 0 : np), a.Wp, w.xx, w.yy, w.dx, w.dy);
          if (b.tflags) {  // the four corners can straddle up to four image tiles
            b.tflags[(w.yy / kTileY) * b.tflags_tiles_x + w.xx / kTileX] = 1;
            b.tflags[(w.yy / kTileY) * b.tflags_tiles_x + (w.xx + 1) / kTileX] = 1;
            b.tflags[((w.yy + 1) / kTileY) * b.tflags_tiles_x + w.xx / kTileX] = 1;
            b.tflags[((w.yy + 1) / kTileY) * b.tflags_tiles_x + (w.xx + 1) / kTileX] = 1;
          }
          nfall++;
        }
      }
    }
  }
  if (nfall) atomicAdd(&sfall, nfall);
  __syncthreads();
  if (tid == 0 && sfall) atomicAdd(b.fallback, sfall);
  if (has_win) {
    const size_t plane_off = c.plane ? np : 0;
    constexpr int kCells = kBinWindow * kBinWindow / 256;  // (see fe_splat_lds_kernel: reads first, then the flush)
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
        const size_t at = plane_off + (size_t)(c.wy0 + ly) * a.Wp + (c.wx0 + lx);
        if (FIXED) atomicAdd(b.fixed + at, v);
        else atomic_add_f32(a.planes + at, (float)((double)v * kFixInv));
        if (b.tflags) b.tflags[((c.wy0 + ly) / kTileY) * b.tflags_tiles_x + (c.wx0 + lx) / kTileX] = 1;
      }
    }
  }
}
void launch_be_splat_lds(const BeSplatArgs &a, const BinnedEvents &b, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
  if (b.nchunks <= 0) return;
  if (b.fixed) {
    if (t0 || t1) hipExtLaunchKernelGGL(be_splat_lds_kernel<true>, dim3(b.nchunks), dim3(256), 0, s, t0, t1, 0, a, b);
    else hipLaunchKernelGGL(be_splat_lds_kernel<true>, dim3(b.nchunks), dim3(256), 0, s, a, b);
  } else {
    if (t0 || t1) hipExtLaunchKernelGGL(be_splat_lds_kernel<false>, dim3(b.nchunks), dim3(256), 0, s, t0, t1, 0, a, b);
    else hipLaunchKernelGGL(be_splat_lds_kernel<false>, dim3(b.nchunks), dim3(256), 0, s, a, b);
  }
}

}  // namespace cmx
