// cmx_internal.hpp -- kernel parameter blocks and launcher prototypes shared by cmx_kernels.hip, cmx_binning.hip and
// the host side of the C ABI (cmx_context.hpp).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include "cmx_frcg_sm.hpp"
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
  const double *lut2;      // W*H*2 (x, y), 16-byte entries, present when every z == 1 (image_geometry's rays); else null
  float *planes;           // [1 + 3][H][W]: IWE, dI/dwx, dI/dwy, dI/dwz
  // device-driven solve (cmx_chain.cpp): omega is read from device memory -- written there by the previous evaluation's
  // finalize step -- instead of (wx, wy, wz); and the kernel returns at once when *skip != 0 (the solve has finished)
  const double *w_dev;
  const int *skip;
};

// per event batch, written by the pose-table kernel.  The rotation lives in its own dense table (72 B per batch):
// the tile-ordered splat reads it at random batch indices, and 3.6 MB for 50k batches stays L2-resident where the
// 216-byte combined record did not (238 MB fetched per 5M-event launch, profiles/r01c_be_fastpath.txt).
struct PoseR { double R[9]; };  // so3.matrix(), row-major
struct PoseEntry {
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
  const PoseR *poseR;
  const PoseEntry *poses;
  const double *lut;
  const double *lut2;      // as in FeSplatArgs
  float *planes;  // [2 + P][Hp][Wp]: IL_old, IL_new, derivative planes
};

// Device-driven solve: the finalize step of an evaluation advances the FR-CG state machine (cmx_frcg_sm.hpp) in device memory
// with the cost / gradient it has just reduced, writes the NEXT evaluation point to x_req and decides whether the gradient pass
// queued behind it runs (FinalizeArgs::gate_out) -- the next evaluation's kernels, already queued by the host, start without a
// round trip to it.  The host replays the same machine on what the result blocks report and takes over on any disagreement.
constexpr int kTailShards = 8;   // ticket counters sharded by blockIdx % 8 (the XCD of a workgroup, for speed only)
constexpr int kTailStride = 32;  // counters 128 B apart; [kTailShards] shard counters, then the top counter
constexpr int kChainMaxN = 3;  // parameters of a device-driven solve (front end)
typedef FrcgSMFix<kChainMaxN> ChainMachine;
struct ChainDev {  // one device allocation, initialised by one copy
  ChainMachine sm;
  double x_req[kChainMaxN];  // evaluation point of the next slot (read by its splat / gather)
  int done;                  // set once the machine has finished: every later kernel of the chain returns at once
  int abort_flag;            // the workgroups of a self-gating gradient pass and the machine disagreed (never expected)
  // self-gating slots: the image pass adds its tiles' two moments (sum B, sum B^2) to 8 shard rows of buffer (slot & 1) with
  // fp64 atomics instead of handing a table to a finalize of its own; rows 128 B apart
  double macc[2][kTailShards][16];
  // self-gating slots: the acceptance test of the slot with parity p, published by the finalize of the slot BEFORE it.  The
  // workgroups of a self-gating launch read gate[p]; the finalize of that very launch (which may run in workgroup 0 while
  // others have not started: cost-only outcome) writes gate[p ^ 1] -- never the words the launch itself is still reading
  // (the gate used to be read from `sm`, which that finalize rewrites).
  struct Gate { double thr; int mode; int pad; } gate[2];
};
struct ChainArgs {
  ChainMachine *sm;  // the machine in device memory (copied to LDS and back by the finalize that advances it); null = off
  double *x_req;
  int *done;
  int stage;         // 0: this finalize ends a cost evaluation, 1: a gradient pass, 2: a self-gating slot (cost, then the gradient
                     //    if the launch computed one: FinalizeArgs::gP > 0)
  int *abort_flag;
  const ChainDev::Gate *gate_cur;  // self-gating slots: the test this launch's workgroups evaluate (ChainDev::gate[parity]) ...
  ChainDev::Gate *gate_next;       // ... and where its finalize publishes the next slot's
  const ChainMachine *sm_src;  // first slot of a warm-started solve: the machine's INITIAL state, read from pinned host memory by the
                               // one finalizing workgroup (no copy in front of the solve); the slot's kernels get omega as arguments,
                               // no end-of-solve flag, and take the machine's first request for what it always is (cost + gradient)
};
constexpr int kChainExtra = 3 + kChainMaxN;  // result words appended by a chained finalize: need-gradient flag, phase, done, next point

struct FinalizeArgs {
  int P, nblk, measure;
  double npix;
  const double *partials;  // [2+2P][nblk]
  double *sums;            // [2+2P] device scratch
  double *result;          // mapped host: [0]=contrast, [1]=mean, [2..2+P) = gradient
  // adjoint mode: gradient = (2/N) * sum over blocks of gpartials[b][k]
  const double *gpartials;
  int gblocks, gP;
  unsigned *fallback;      // LDS-splat fallback counter: copied to result[4094] and reset (may be null)
  int direct;              // 1: sum rows 0,1 of `partials` inside finalize (no reduce_partials launch)
  const unsigned *nvalid;  // direct mode: device count of valid entries per row (tile work list); null = nblk
  int mu_free;             // 1: gpartials rows hold [S1 (gP) | S2 (gP)], grad = (2/N)(S1 - mu*S2)
  unsigned long long ticket;  // written after the results to result[kTicketSlot]: the host polls it
  // tail finalize of the back-end gradient: instead of a [column][workgroup] table the per-batch pass adds its column sums to
  // kTailShards rows of accumulators (device-scope fp64 atomics, row = workgroup % kTailShards); the finalize sums the rows
  // and stores zeros back, so that the buffer is all-zero between launches.  Null = the table (gpartials / gblocks)
  double *gacc;
  int gacc_stride;            // doubles per shard row (>= number of columns)
  // gated gradient pass (cmx_hint_next_df): the finalize of a cost-only evaluation decides from its own cost f = -contrast
  // whether the gradient pass queued behind it runs: *gate_out = (f < thr | f <= thr | !(f >= thr) | 1) for mode 1..4
  int *gate_out;
  double gate_thr;
  int gate_mode;
  ChainArgs chain;
  // self-gating slots of the device-driven solve: image moments from accumulator rows (ChainDev::macc) instead of a table; the
  // finalize zeroes the OTHER buffer (the next slot's); nout_pad: result words before the chain's extension whatever gP is
  const double *macc;
  double *macc_clear;
  int nout_pad;
};
// contrast from the two image moments: the one expression shared by the finalize step and by the workgroups of a self-gating
// gradient pass (they must take the machine's decision from bitwise the same number)
static inline __host__ __device__ double contrast_from_sums(double s0, double s1, double N, int measure, double *mu_out) {
  const double mu = s0 / N;
  *mu_out = mu;
  if (measure == 1) return s1 / N;
  double var = s1 / N - mu * mu;
  if (var < 0) var = 0;
  const double sd = sqrt(var);
  return sd * sd;
}
// the condition of the gate, the same expression on the device (finalize) and on the host (which result to expect)
static inline __host__ __device__ int gate_condition(double contrast, double thr, int mode) {
  const double f = -contrast;
  return mode == 1 ? (f < thr) : mode == 2 ? (f <= thr) : mode == 3 ? !(f >= thr) : 1;
}
// tail of the mapped result buffer (doubles / u64 bit patterns)
constexpr int kChecksumSlot = 4092;  // xor of the bit patterns of result[0..nout) and result[kFallbackSlot], ^ ticket*kTicketMix
constexpr int kTicketSlot = 4093;    // ticket of the last finished evaluation
constexpr int kFallbackSlot = 4094;  // votes that left their LDS window in that evaluation
constexpr int kAlphaSlot = 4095;     // alpha mirror (back end)
constexpr int kXsetSlot = 4088;      // sharded panoramas (xset_kernel): [0] tiles in the next evaluation's exchange set, [1] flagged tiles
                                     // this evaluation's exchange did not cover, [2] flagged tiles, [3] = stamp:
                                     // bits[0] ^ bits[1] ^ bits[2] ^ (launch sequence number * kTicketMix)
constexpr unsigned long long kTicketMix = 0x9E3779B97F4A7C15ull;


// Tail finalize (cmx_kernels.hip, tail_arrive): the LAST kernel of an evaluation -- image_moments / image_adjoint2 (cost-only),
// fe_gather / be_gather4 with the per-batch pass folded in / be_gather_batch (adjoint gradient) -- runs the finalize step in
// its last-arriving workgroup, so an evaluation ends without the one-workgroup finalize launch and the boundary in front of it.
constexpr int kTailCounterWords = (kTailShards + 1) * kTailStride;
constexpr int kGaccStride = 2 * 3 * kMaxKnots + 2;  // doubles per accumulator row: S1 | S2 columns (+ spare), see FinalizeArgs::gacc
struct TailArgs {
  unsigned *counters;  // all-zero between launches (the last arrivers reset what they completed); null = no tail finalize
  FinalizeArgs fin;
  int poll;            // 1 (front-end plain gather, round 6): workgroup 0 finalizes once every other workgroup has ARRIVED -- fire-and-forget
                       // arrival atomics on counters[0], one polling lane -- instead of the last arriver found through two levels of
                       // returning ticket atomics (two dependent memory trips less at the end of every gradient evaluation)
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
  float *zero_ptr;      // optional: the OTHER accumulation buffer (previous evaluation's planes); every workgroup clears
  int zero_planes;      // its own tile there, so the next evaluation needs no memset launch (ping-pong accumulation)
  // tile occupancy (one byte per kTileX x kTileY tile; back end, LDS splat): a panorama is mostly empty, so the image
  // passes skip every tile with no vote (and no global-map content) within the filter's reach
  const unsigned char *flags_cur;   // tiles of src_a/src_b that received votes in this evaluation; null = all tiles
  const unsigned char *flags_igp;   // tiles where igp is non-zero (only read when igp and flags_cur are set)
  unsigned char *flags_other;       // tiles of zero_ptr that are dirty: cleared (and un-flagged) selectively; null = all
  int tiles_y;
  // large panoramas (> kTileListMin tiles): a one-workgroup pre-pass (launch_tile_list) compacts the tiles that need
  // work into tile_list (bit 31: active = run the filters, bit 30: dirty = clear the partner's tile; low bits: tile) and
  // the image kernels walk that list with a bounded grid instead of launching one workgroup per panorama tile
  const unsigned *tile_list;
  const unsigned *tile_count;
  const int *skip;      // optional (device-driven solve): the launch does nothing when *skip != 0
  double *macc;         // optional (device-driven solve, self-gating slots): moment accumulator rows instead of `partials`
  TailArgs tail;        // image_moments only (cost-only evaluations, P == 0): finalize in the last-arriving workgroup
};
constexpr int kTileListMin = 2048, kTileListGrid = 1024;
// reach: pixels of filter support beyond the tile (r for the moments pass, 2r for the adjoint pass)
void launch_tile_list(const ImgArgs &a, int reach, unsigned *list, unsigned *count, unsigned *next_count, bool ordered, hipStream_t s);

// fused image pass of the adjoint gradient: B = G*A (moments of B), Jt = G^T B^ in ONE kernel.
// G^T(B - mu) = G^T B - mu*c with c = G^T 1 = cx(x)*cy(y) (1 in the interior, differs only within r of the border), and the
// bilinear-derivative weights sum to zero, so the mu term only matters for votes next to the border: the gather
// accumulates it separately (S2) and finalize applies  grad = (2/N) (S1 - mu*S2).
struct ImgAdjArgs {
  ImgArgs img;      // composition + taps + partials (rows 0,1) ; P / dplanes / out_blurd unused
  float *jt;        // out: G^T B^  (W*H)
  const float *Mx, *My;  // optional: banded G^T G per axis, [L][4r+1]: the three-phase form of the pass (r == 4)
};
size_t image_adjoint_lds_bytes(int r);
int image_adjoint_tiles_x(int W);
int image_adjoint_tiles(int W, int H);
void launch_image_adjoint(const ImgAdjArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

struct FeGatherArgs {
  FeSplatArgs ev;          // same event / camera description as the splat
  const float *itilde;
  double *gpartials;       // [nblocks][3]
  const uint32_t *sxy;     // optional: events in destination-tile order (better LUT / Itilde locality) ...
  const uint32_t *sbatch;  // ... with their batch indices; null = time order
  const double *sb, *sdt;  // optional, with sxy: per-event bearing (x, y) and dt in the same order (see BinnedEvents)
  const double *tb;        // optional, time order (sxy == null): per-event bearing (x, y) stream
  const float *cx, *cy;    // G^T 1 factors (W and H floats) when itilde holds G^T B (mu-free form); null: itilde = G^T(B-mu)
  int r;                   // blur radius (defines the border band where cx, cy differ from 1)
  const int *gate;         // optional: the launch does nothing when *gate == 0 (gated gradient pass, FinalizeArgs::gate_out)
  TailArgs tail;           // finalize in the last-arriving workgroup (counters == null: separate finalize launch)
};

struct BeGatherArgs {
  BeSplatArgs ev;
  const float *itilde;
  int P;                   // 3 * (K - num_fixed)
  double *gpartials;       // [2P][be_batch_blocks(nb)]: S1 columns then S2 columns
  const float *cx, *cy;    // as in FeGatherArgs
  int r;
  double *vparts;          // [nb][parts_per_batch][6]: per-batch partial sums of V (3) and of the border vector U (3)
  int parts_per_batch;     // wave slices a batch can touch
  int slice_shift;         // log2 of the events one wave pass covers: 6 (one event per lane) or 8 (four per lane)
  int deterministic;       // per-parameter block sums in a fixed order instead of LDS fp64 atomics
  const double *tb;        // optional: bearing (x, y) of every event in TIME order (16 B, z == 1): coalesced stream for the
                           // four-events-per-lane pass instead of four divergent bearing-table gathers per lane
  const int *gate;         // optional: the launch does nothing when *gate == 0 (gated gradient pass, FinalizeArgs::gate_out)
  TailArgs tail;           // be_gather_batch / folded be_gather4: finalize in the last-arriving workgroup (counters == null: separate
                           // launch); tail.fin.gacc set without counters: accumulator rows only (sharded split evaluation)
  int fold;                // 1: fold the per-batch pass into be_gather4 when the launcher's conditions hold (be_gather_folds)
};
bool be_gather_folds(const BeGatherArgs &a);

struct AlphaArgs {
  const float *igp, *il_old, *il_new;
  int npix;
  double *partials;  // [5][nblk]
  int nblk;
  double *alpha;     // device scalar out
  double *result_alpha;  // mapped host copy
};

// ---- LDS-privatised splat (events sorted by destination tile once per packet / window) ----------------
constexpr int kBinTile = 32;     // destination tile edge (pixels)
constexpr int kBinMargin = 16;   // window = tile + margin on every side: 64 x 64 fp32 per plane in LDS
constexpr int kBinWindow = kBinTile + 2 * kBinMargin;
constexpr int kBinStride = 67;   // LDS row stride of a window (floats): odd, with stride+-1 poor in factors of 2, so votes along
                                 // vertical / diagonal edges spread over the 32 LDS banks instead of piling onto one

struct Chunk { int wx0, wy0, beg, end, plane, tile; };  // LDS window origin (pixels), sorted-event range, target plane
                                                         // (back end: 0 = IL_old, 1 = IL_new); wx0 < -1e8: no window;
                                                         // tile: the chunk's sort tile (-1: the no-window sentinel)

// Tile-dataflow fusion of the adjoint image pass into the front-end LDS splat (round 6; fe_splat_lds_kernel<.., FUSE>,
// cmx_tilepass.hpp).  The image tiles of the fused pass ARE the 32 x 32 sort tiles.  A chunk's votes stay inside its window
// (tile + 16 px) or, on the global path, within kFuseReach = 56 px of its tile, and the pass of tile T reads the raw image on
// T + 2r = 8 px: everything T needs comes from the chunks of T's 5 x 5 tile neighbourhood.  Every chunk workgroup, once its
// window has been flushed, arrives on the counters of the (up to) 25 tiles around its own.  The launch carries one more workgroup per image tile behind the chunk workgroups: it waits for
// its tile's count and runs B = G I (moments), Jt = G^T G I for that tile -- no kernel boundary between the splat and the image
// pass, no second launch, and the 300 tile passes of a 640 x 480 image run side by side the moment the last chunks land
// (the first form, "the completing arrival runs the pass", put up to nine passes in a row on the late finishers: splat + image
// 50 us instead of 16, profiles/r06_fused_ab.txt).  Votes beyond kFuseReach are not covered by the counts: an evaluation that
// reports any (kFuseUnsafe) is repeated by the host after a fresh sort (cmx_frontend.cpp).
constexpr int kFuseCntStride = 32;  // words between two tiles' arrival counters: one 128-byte line each -- 300 polling lanes on ten
                                    // shared lines serialised at the memory side and held up the chunks' own atomics (chunk phase
                                    // 7 -> 13 us, profiles/r06_fused_ab.txt)
// Votes on the global-atomic path (outside the chunk's 64 x 64 LDS window) are still covered by the tiles' arrival counts as long
// as they land within kFuseReach pixels of the chunk's own tile: the chunk arrives on the 5 x 5 tiles around its own, and a tile's
// pass reads tile + 2r = 8 px -- 2 * 32 - 8 = 56 px of reach (3.8 rad/s of omega at the far end of a 50 ms packet).  A vote beyond
// that raises kFuseUnsafe in the fallback word; a tile workgroup that gave up waiting raises kFuseIncomplete.  The low 30 bits stay
// the count of global-path votes (the 3 % re-sort rule of the LDS splat).
constexpr int kFuseNbr = 2;  // tiles on every side of a chunk's tile it arrives on
constexpr int kFuseStrips = 1;  // image-pass workgroups per 32 x 32 tile (strips of 32 x 32 / kFuseStrips lines, each with its own arrival
                                // counter and moment row).  2 measured: no gain -- the 300 passes are bound by the CUs' fp64-conversion / LDS
                                // throughput, not by one pass's latency (profiles/r06_fused_ab.txt)
constexpr int kFuseReach = kFuseNbr * 32 - 8;
constexpr unsigned kFuseUnsafe = 0x80000000u, kFuseIncomplete = 0x40000000u, kFuseCountMask = 0x3fffffffu;
struct FusedArgs {
  int tiles_x, tiles_y;         // sort-tile grid (kBinTile pixels)
  const int *nbr_expected;      // [tiles * kFuseStrips]: chunks in the tile's 5 x 5 neighbourhood (0: no vote can reach it: nobody runs it)
  unsigned *nbr_cnt;            // [tiles * kFuseStrips * kFuseCntStride]: arrivals so far; all-zero between launches (the strip's
                                // workgroup stores 0)
  float taps[9];                // radius 4 only
  const float *Mx, *My;         // banded G^T G per axis (cmx_context.cpp upload_gt1)
  float *jt;                    // out: G^T G I
  float *zero_ptr;              // the OTHER accumulation buffer: a tile's pass clears its own tile there (ping-pong); may be null
  double *partials;             // [2][tiles * kFuseStrips]: per-strip sum B, sum B^2 (rows of inactive tiles stay zero: written at sort time)
  double *macc;                 // device-driven solve: moment accumulator rows (ChainDev::macc) instead of `partials`; else null
  unsigned long long *trace;    // diagnostics (env CMX_FUSE_TRACE): [workgroup][8] wall-clock stamps -- start, inputs complete / chunk
                                // flushed, end, role, then the tile pass's phases (raw loaded, row pass done, column pass done); null = off
  int debug;                    // diagnostics (env CMX_FUSE_DEBUG, timing experiments only -- results are WRONG when set): 1 tile
                                // workgroups leave at once, 2 they wait but skip the pass, 4 chunks do not arrive (with 1)
  // ---- ONE-LAUNCH evaluation (fe_splat_lds_kernel<.., FUSE = 2>): the gradient gather and the finalize step ride in the same
  // launch as well.  Behind the strip workgroups come `gather_blocks` GATHER workgroups: each loads and warps its slice of the
  // tile-sorted events while the splat is still running (streams, fp64 warp, Jacobian rows: nothing of that depends on Jt), marks
  // the tiles its vote cells lie in, waits for exactly those tiles' passes (tile_done[t] == seq), reads the 4 Jt cells per event
  // (agent-scope loads: the strip workgroups store Jt write-through), adds its six sums to the accumulator rows and arrives; the
  // last arriver waits for every strip (tiles_done == *n_active) and runs the finalize: contrast, gradient, fallback word,
  // checksum + ticket to the mapped result block.  Every wait points at workgroups with LOWER indices that wait for nobody
  // further up, and is bounded (kFuseIncomplete).
  int gather_blocks;            // 0: two-launch form (FUSE = 1)
  int self_serve;               // 1: the self-service one-launch form (FUSE = 3, cmx_selfserve.hpp): no strip / gather workgroups at all
  int gather_per_block;         // events per gather workgroup (a multiple of the workgroup size; at most 4 per thread)
  unsigned seq;                 // this launch's stamp in tile_done[]
  unsigned *tile_done;          // [tiles * kFuseStrips * kFuseCntStride]: == seq once the strip's Jt and moment row have been stored
  unsigned *tiles_done;         // strips finished in this launch; reset by the finalizing workgroup
  const int *n_active;          // strips that run (nbr_expected > 0), written with the chunk table
  const float *cx, *cy;         // G^T 1 factors (border band of the mu term)
  double *gacc;                 // kTailShards accumulator rows (FinalizeArgs::gacc), all-zero between launches
  int gacc_stride;
  unsigned *tail_counters;      // last-arriver tickets of the gather workgroups (TailArgs::counters)
  double *result;               // mapped host result block
  unsigned long long ticket;
  double npix;
  int measure;
};

struct BinnedEvents {
  const uint32_t *sxy;     // packed events in sorted order
  const uint32_t *sbatch;  // batch index of each sorted event
  const Chunk *chunks;
  int nchunks;             // launch bound: workgroups beyond *nchunks_dev (the table's true length) return at once
  const int *nchunks_dev;
  unsigned *fallback;      // events that left their window and took the global-atomic path (device counter)
  unsigned char *tflags;   // optional: image-tile occupancy map marked by every vote that reaches global memory
  int tflags_tiles_x;
  const double *sb;        // front end, optional: bearing (x, y) of each sorted event (16 B, z == 1) ...
  const double *sdt;       // ... and its batch's dt: coalesced streams instead of two divergent table gathers per event
  unsigned long long *fixed;  // deterministic mode: 2^-30 fixed-point planes every global vote is added to (else nullptr)
  int sort_tiles_x, sort_tiles_y;  // front end: the sort-tile grid (global-path votes are classified by reach, see kFuseReach)
};

// binning: key = destination tile under the current parameters (ntiles = "not accepted right now")
void launch_fe_bin_keys(const FeSplatArgs &a, int tiles_x, int ntiles, uint32_t *keys, uint32_t *idx, hipStream_t s);
void launch_be_bin_keys(const BeSplatArgs &a, int tiles_x, int ntiles, uint32_t *keys, uint32_t *idx, hipStream_t s);
int sort_pairs_u32(void *temp, size_t *temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                   uint32_t *vout, unsigned n, int end_bit, hipStream_t s);  // rocprim radix sort (library op)
void launch_apply_perm(const uint32_t *xy, const uint32_t *idx_sorted, int per_batch, int n, uint32_t *sxy,
                       uint32_t *sbatch, hipStream_t s);
void launch_tile_lower_bound(const uint32_t *keys_sorted, int n, int ntiles_plus2, int *tile_start, hipStream_t s);
// counting sort by destination tile (cmx_binning.hip): the default whenever the key space fits an LDS histogram
bool count_sort_ok(int nbins);
size_t count_sort_scratch_ints(int n, int nbins);
void launch_count_sort(const FeSplatArgs *fe, const BeSplatArgs *be, int tiles_x, int ntiles_img, const uint32_t *xy,
                       int per_batch, int n, uint32_t *keys, int *scratch, int *tile_start, uint32_t *sxy, uint32_t *sbatch,
                       double *sb, double *sdt, hipStream_t s);
// t0 / t1 (optional): events bracketing exactly the kernel(s) of the launch (hipExtLaunchKernelGGL start / stop events,
// the timestamps rocprofv3 reports) for the live roofline measurement of bench.py
void launch_fe_splat_lds(const FeSplatArgs &a, const BinnedEvents &b, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr,
                         const FusedArgs *fused = nullptr);  // fused: the image pass runs inside this launch (see FusedArgs)
void launch_be_splat_lds(const BeSplatArgs &a, const BinnedEvents &b, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// deterministic mode: fixed-point planes -> fp32 planes (non-zero entries only; `planes` is all-zero before), fixed := 0
// bearing (x, y) of every event in time order (back-end gather stream)
void launch_bearing_stream(const uint32_t *xy, const double *lut2, int W, int n, double *tb, hipStream_t s);
void launch_fixed_to_float(unsigned long long *fixed, float *planes, size_t n, hipStream_t s);

void launch_fe_splat(const FeSplatArgs &a, bool deriv, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_be_pose_table(const SplineArgs &spline, const long long *d_batch_t, int nb, int order, bool want_j,
                          PoseR *outR, PoseEntry *out, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_be_splat(const BeSplatArgs &a, bool deriv, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_image_moments(const ImgArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_finalize(const FinalizeArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_xset(const unsigned char *flags, int tiles_x, int tiles_y, const unsigned char *cur_member, int *next_list,
                 unsigned char *next_member, int *miss_list, double *out, unsigned long long seq, hipStream_t s);
struct XsetPeers { const float *p[16]; int n; int xdev; };  // xdev: the buffers live on several devices (system-scope acquire first)  // the members' packed send buffers of one exchange (cmx_group.cpp's direct transport)
void launch_xset_sum_unpack(const XsetPeers &in, float *planes, size_t np, int W, int H, const int *list, int n, unsigned char *flags,
                            int ntiles, hipStream_t s);
void launch_xset_copy(bool unpack, float *planes, size_t np, int W, int H, const int *list, int n, float *stage, unsigned char *flags,
                      int ntiles, hipStream_t s);
void launch_tile_flags_pair(const float *a, const float *b, int W, int H, unsigned char *flags, hipStream_t s);
void launch_tile_flags(const float *plane, int W, int H, unsigned char *flags, hipStream_t s);  // flags[tile] = 1 where plane != 0
void launch_alpha(const AlphaArgs &a, hipStream_t s);
void launch_reduce_partials(const FinalizeArgs &a, hipStream_t s);
void launch_reduce_gpartials(const double *gpartials, int gblocks, int P, double *gsum, hipStream_t s);
void launch_finalize_only(const FinalizeArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
int launch_fe_gather(const FeGatherArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);  // returns the number of blocks (rows of gpartials)
int launch_be_gather(const BeGatherArgs &a, int nb, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, hipEvent_t b0 = nullptr,
                     hipEvent_t b1 = nullptr);  // returns the rows of gpartials (batch-kernel blocks)
int be_batch_blocks(int nb);
int gather_blocks(int n);
int fe_gather_blocks(int n);
int fe_selfserve_capacity();  // workgroups of the self-service one-launch form that are resident at once on the current device
// contrast_ImageGradientMagnitude (front end, contrast_measure = 2): Sobel moments of the blurred planes
struct SobelArgs {
  int W, H, P;
  const float *planes;   // [1+P][H][W] blurred I, D_0..D_{P-1}
  double *partials;      // [1+P][nblk]: sum(gx^2+gy^2), sum(gx*dgx_k + gy*dgy_k)
  int nblk;
};
void launch_sobel_moments(const SobelArgs &a, hipStream_t s);
int sobel_blocks(int W, int H);

// back-end window cut from the device-resident event store: sub-sampling restarts per batch, old/new flag from the timestamps
// chunk table built on the device from the tile offsets (no host round trip): tile_start[ntiles+2] -> chunks, *count
// fused (front end, optional): nbr_expected / nbr_cnt / partials of FusedArgs are (re)initialised for this table
struct FusedTables { int tiles_y; int *nbr_expected; unsigned *nbr_cnt; double *partials; int *n_active; unsigned *tiles_done; };
bool fused_tables_ok(int ntiles, int planes_per_tile);  // the chunk-table kernel can build them for this tile grid
void launch_build_chunks(const int *tile_start, int ntiles, int planes_per_tile, int tiles_x, int margin, int M, Chunk *chunks,
                         int *count, unsigned long long *count_host, unsigned binning_id, hipStream_t s,
                         const FusedTables *fused = nullptr);
void launch_be_batch_times(const long long *t, long long n, int B, int nb, long long start_ns, long long dt_ns, int order, int K,
                           long long *bt, long long *err, hipStream_t s);
void launch_be_pack_from_store(const uint32_t *raw, const long long *t, long long n, int B, int rate, int per_batch,
                               int n_packed, long long t_next, uint32_t *out, hipStream_t s);

// global-map upkeep (once per window)
void launch_update_map(float *IG, const float *IL_old, const unsigned char *visits, int npix, int max_update_times,
                       hipStream_t s);
void launch_mark_visited(const BeSplatArgs &cam, const double R[9], int sensor_h, int radius, unsigned char *mask,
                         unsigned char *visits, hipStream_t s);
void launch_interleave3(const float *planes, float *out, int npix, hipStream_t s);
size_t image_lds_bytes(int r);

}  // namespace cmx
