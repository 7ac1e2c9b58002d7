// cmx_context.hpp -- the evaluator context behind include/cmax_hip.h's opaque cmx_ctx / cmx_events, and the helpers
// the translation units of the C ABI share:
//   cmx_context.cpp   context life cycle, options, timing, error text, event validation
//   cmx_pipeline.cpp  one evaluation on the stream: accumulate buffers, tile sort, image passes, gather, finalize, ticket
//   cmx_frontend.cpp  cmx_frontend_*      cmx_backend.cpp  cmx_backend_*      cmx_events.cpp  cmx_events_*
//   cmx_comm.cpp      RCCL communicator inside the evaluator (sharded evaluations)
// Internal to libcmaxhip.so; gfx950 only.  There is NO CPU fallback: every entry point fails loudly without HIP.
#pragma once
#include "../../include/cmax_hip.h"
#include "../../include/cmax_hip_diag.h"  // the A/B option keys: internal + tests, not the host surface

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

#include "cmx_hostpool.hpp"
#include "cmx_internal.hpp"

using namespace cmx;  // internal header of one library: the parameter blocks of cmx_internal.hpp are used unqualified

enum { KIND_FE = 1, KIND_BE = 2 };
constexpr long long kMaxPixels = 1LL << 29;  // per plane (sensor or panorama): pixel loops use 32-bit ints, up to 3 planes interleaved
constexpr long long kMaxEvents = 1LL << 30;  // kernels index events with 32-bit ints (grid-stride loops add up to 2^19)
// Re-sort the events by destination tile once more than this share of the votes left their LDS windows: a vote on the
// global-atomic path costs the 1M-event splat ~3.7 us per percent (9.5 % -> 44 us instead of 9.5), a re-sort ~60 us once
constexpr double kRebinFallbackFrac = 0.03;
// the fallback word of a result block (kFallbackSlot): count of global-path votes in the low 30 bits, kFuseUnsafe / kFuseIncomplete above
inline unsigned fallback_word(double slot) { return slot >= 0.0 && slot < 4294967296.0 ? (unsigned)slot : 0u; }
inline double fallback_count(double slot) { return (double)(fallback_word(slot) & cmx::kFuseCountMask); }
inline unsigned fallback_flags(double slot) { return fallback_word(slot) & ~cmx::kFuseCountMask; }

struct TimedSpan { int cls; hipEvent_t a, b; };
typedef struct ncclComm *ncclComm_t;  // as <rccl/rccl.h> declares it; only cmx_comm.cpp includes that header

struct cmx_group;  // cmx_group.cpp: one-process multi-GPU group (members, worker threads, transport)

struct cmx_ctx {
  int kind = 0, device = 0;
  cmx_group *group = nullptr;  // set on every member of a group; the member with group_rank 0 is the handle the caller holds
  int group_rank = 0;
  int sched_class = 0;  // cmx_set_sched_class: -1 background (yields to urgent contexts of its device), 0 normal, 1 urgent
  bool group_partial_grad = false;  // the last evaluation's gradient covers this member's events only (cmx_comm.cpp: finish_exchanged)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // sensor + LUT
  int W = 0, H = 0;
  double *d_lut = nullptr;
  long long *d_batch_err = nullptr;  // error kind / position reported by the device-side batch-time pass
  double *d_lut2 = nullptr;  // (x, y) pairs, 16-byte entries: present when the caller's table has z == 1 everywhere

  // packed events
  uint32_t *d_xy = nullptr;
  size_t xy_cap = 0;
  uint32_t *h_xy = nullptr;  // pinned staging for the packed events (host packing runs on several threads)
  size_t h_xy_cap = 0;
  double *h_dts = nullptr;   // pinned staging for the per-batch dt table (uploaded asynchronously with the events)
  size_t h_dts_cap = 0;
  double blur_sigma_built = -1.0;  // sigma the blur taps / G^T 1 factors / operator tables on the device were built for (-1: none)
  int n_packed = 0, per_batch = 1, nb = 0;
  bool have_data = false;

  // config shared by both ends
  int batch = 100, sample_rate = 1, measure = CMX_VARIANCE;
  double sigma = 0;
  int radius = 0;
  float taps[2 * kMaxRadius + 1] = {1.f};
  int grad_mode = CMX_GRAD_ADJOINT, splat_mode = 1;  // production configuration by default; cmx_set_option selects the reference-shaped path

  // front end
  double fx = 0, fy = 0, cx = 0, cy = 0;
  double *d_batch_dt = nullptr;
  size_t batch_cap = 0;

  // back end
  int Wp = 0, Hp = 0, order = 0, K = 0, num_fixed = 0;
  long long *d_batch_t = nullptr;
  PoseEntry *d_poses = nullptr;
  PoseR *d_poseR = nullptr;
  size_t batch_t_cap = 0, poses_cap = 0, poseR_cap = 0;
  SplineArgs *h_spline = nullptr;  // temp-trajectory description, passed to the pose-table kernel by value
  std::vector<Quat> knots0;
  float *d_IG = nullptr, *d_IGp = nullptr;
  unsigned char *d_visits = nullptr, *d_mask = nullptr;  // IG_update_times_map_ and the per-pose scratch mask
  bool ig_nonzero = false, first_iter = true;
  double *d_alpha = nullptr;

  // image planes
  int imgW = 0, imgH = 0;  // W,H (front end) or Wp,Hp (back end)
  float *d_accum = nullptr;
  size_t accum_cap = 0, accum_count = 0;
  // ping-pong partner of d_accum (fast path, context-owned memory only): the image kernel of evaluation k clears the
  // buffer evaluation k-1 used, so evaluation k+1 splats into it without a memset launch
  float *d_accum_alt = nullptr;
  size_t accum_alt_cap = 0;
  bool accum_clean = false, alt_clean = false;  // buffer known to be all-zero over the planes the fast path uses
  int pingpong_planes = 0;                     // planes being ping-ponged by the pending evaluation (0 = off)
  bool accum_external = false;
  float *d_scratch = nullptr;  // blurred-plane readback scratch
  size_t scratch_cap = 0;
  int last_P = 0;              // derivative planes produced by the last accumulate()
  bool accumulated = false;

  // adjoint-gradient scratch: Jt plane, per-block gradient partials
  float *d_itilde = nullptr;  // Jt = G^T (G I)
  size_t itilde_cap = 0;
  float *d_cx = nullptr, *d_cy = nullptr;  // G^T 1 = cx(x)*cy(y): column sums of the REFLECT_101 blur operator
  size_t cx_cap = 0, cy_cap = 0;
  float *d_Mx = nullptr, *d_My = nullptr;  // banded G^T G per axis, [L][4r+1] (composite image pass)
  size_t Mx_cap = 0, My_cap = 0;
  int Mx_radius = -1;                      // blur radius the tables were built for (-1: none)
  bool composite_image = true;             // CMX_OPT_COMPOSITE_IMAGE
  bool fold_batch = true;                  // CMX_OPT_FOLD_BATCH
  bool shard_acc = false;                  // set by cmx_comm.cpp around a split evaluation it all-reduces itself (see run_adjoint)
  double *d_gpartials = nullptr;
  size_t gpartials_cap = 0;
  double *d_vparts = nullptr;  // back end: per-batch partial V sums of the gather pass
  size_t vparts_cap = 0;
  double *d_gsum = nullptr;   // this rank's partial gradient sums [P] (caller-owned when external: RCCL reduces it in place)
  size_t gsum_cap = 0;
  bool gsum_external = false;
  bool finish_pending = false; // finish_begin ran, finish_end has not
  bool acc_dirty = false;      // a split evaluation added to d_gacc and has not reached the finalize that zeroes it again
  int pending_P = 0;
  bool last_adjoint = false;  // the last accumulate() ran in adjoint mode with a gradient requested
  // back end: image-tile occupancy of the two ping-pong accumulation buffers and of IGp (see ImgArgs::flags_*)
  unsigned char *d_tflags = nullptr, *d_tflags_alt = nullptr, *d_igp_flags = nullptr;
  size_t tflags_cap = 0;          // tiles
  bool accum_flagged = false;     // every non-zero pixel of d_accum lies in a tile flagged in d_tflags
  bool alt_flagged = false;       // the same for d_accum_alt / d_tflags_alt
  bool igp_flags_valid = false;
  unsigned *d_tile_list = nullptr, *d_tile_count = nullptr;  // compacted work list of the image passes (large panoramas)
  size_t tile_list_cap = 0;
  int tile_count_sel = 0;
  bool adj_direct = false;                   // shape of the moment partials the last adjoint image pass produced
  const unsigned *adj_tile_count = nullptr;  // (run_adjoint phase 1 -> phase 2)
  bool x_valid = false;       // plane 0 (and the pose table) hold the accumulation for last_x
  bool jt_valid = false;      // ... and Jt + the moment rows hold the adjoint image pass of those planes (speculative, see
                              // speculative_jt_ok); cleared by every accumulate
  int64_t spec_images = 0, spec_hits = 0;
  int reuse_image = 1;        // df right after f at the same point reuses the image (CMX_OPT_REUSE_IMAGE)
  int64_t reuse_hits = 0;
  double last_x[3 * kMaxKnots] = {0};  // parameters of the last accumulate (the gather pass re-warps the events)

  // LDS-privatised splat (CMX_OPT_SPLAT_MODE = 1): events sorted by destination tile, chunk table
  bool bin_valid = false;
  uint32_t *d_keys = nullptr, *d_keys_s = nullptr, *d_idx = nullptr, *d_idx_s = nullptr, *d_sxy = nullptr, *d_sbatch = nullptr;
  size_t bin_cap = 0;
  bool deterministic = false;            // CMX_OPT_DETERMINISTIC
  unsigned long long *d_fixed = nullptr;  // its fixed-point vote planes (all-zero between evaluations)
  size_t fixed_cap = 0;
  double *d_sb = nullptr, *d_sdt = nullptr;  // front end: bearing (x, y) and dt of every tile-sorted event
  size_t sb_cap = 0, sdt_cap = 0;
  bool streams_valid = false;
  double *d_tb = nullptr;  // back end: bearing (x, y) of every event in time order (gather stream)
  size_t tb_cap = 0;
  bool tb_valid = false;
  int *d_hist = nullptr;  // counting sort scratch: [bin totals | slices x bins prefix table]
  size_t hist_cap = 0;
  void *d_sort_temp = nullptr;
  size_t sort_temp_cap = 0;
  int *d_tile_start = nullptr;
  size_t tile_start_cap = 0;
  Chunk *d_chunks = nullptr;
  size_t chunks_cap = 0;
  int nchunks = 0;          // launch bound of the chunk table (its true length lives in d_nchunks)
  int *d_nchunks = nullptr;
  // mapped host copy of the table's true length, written by build_chunks as (binning id << 32 | length): a count is only
  // accepted with the id of the latest binning, so a stale store of an earlier, uncollected binning cannot be mistaken for it
  unsigned long long *h_nchunks = nullptr, *d_nchunks_host = nullptr;
  unsigned binning_id = 0;
  bool nchunks_exact = false;  // nchunks has been replaced by the table's true length (read back after the first evaluation)
  unsigned *d_fallback = nullptr;
  // tile-dataflow fusion of the adjoint image pass into the front-end LDS splat (FusedArgs, cmx_internal.hpp; CMX_OPT_FUSED_IMAGE)
  bool fused_image = true;
  int *d_fnbr_expected = nullptr;     // per sort tile: chunk arrivals that complete its 3 x 3 neighbourhood (built with the chunk table)
  unsigned *d_fnbr_cnt = nullptr;     // arrival counters, all-zero between launches
  double *d_fpartials = nullptr;      // [2][tiles] moment rows of the fused pass
  size_t fnbr_cap = 0, fcnt_cap = 0, fpartials_cap = 0;
  unsigned *d_ftile_done = nullptr;   // one-launch evaluation: per-strip stamps (== fuse_seq of the launch that computed the strip)
  unsigned *d_ftiles_done = nullptr;  // ... strips finished in the running launch (reset by its finalizing workgroup)
  int *d_fn_active = nullptr;         // ... strips that run under the current chunk table
  size_t fdone_cap = 0;
  unsigned fuse_seq = 0;
  bool fused_full = false;            // CMX_OPT_FUSED_IMAGE 2 (opt-in, measured slower: profiles/r06_fused_ab.txt): gather + finalize ride
                                      // in the splat launch as well
  bool fused_self = false;            // CMX_OPT_FUSED_IMAGE 3: ONE launch of the chunk workgroups alone (cmx_selfserve.hpp)
  int fused_self_strikes = 0;         // evaluations of that form that ran into a bounded wait (three: the context stops using it)
  int64_t fused_self_evals = 0;
  bool fused_full_done = false;       // the pending evaluation is ONE launch: its finalize carries the ticket, nothing is left to queue
  int64_t fused_full_evals = 0;
  unsigned fused_bin_id = 0;          // binning the three tables above were built for (0: none)
  int fused_tiles_x = 0, fused_tiles_y = 0;
  bool fused_done = false;            // the pending evaluation's splat carried the image pass: Jt and d_fpartials are (being) written
  double *fuse_macc = nullptr;        // set by a self-gating slot of the device-driven solve around its fe_accumulate: the fused pass adds
                                      // the tiles' moments to these accumulator rows (ChainDev::macc) instead of writing d_fpartials
  bool adj_fused = false;             // the moment rows of the last adjoint image pass are d_fpartials (else d_partials)
  unsigned votes_bin_id = 0;          // binning under which d_accum's votes were made by an LDS splat (0: some other way)
  unsigned last_fallback_flags = 0;   // kFuseUnsafe / kFuseIncomplete of the last collected evaluation
  bool force_rebin = false;           // a fused evaluation reported votes outside their windows: sort again before the next splat
  int64_t fused_evals = 0, fused_redos = 0, fused_timeouts = 0;
  unsigned long long *d_fuse_trace = nullptr;  // diagnostics (env CMX_FUSE_TRACE = output file): see FusedArgs::trace
  size_t fuse_trace_cap = 0, fuse_trace_n = 0;
  int64_t rebin_count = 0;
  double last_fallback_frac = 0;
  bool last_used_lds = false;
  bool fallback_pending = false;  // an LDS splat ran since the counter was last read back

  // reductions
  double *d_partials = nullptr, *d_sums = nullptr;
  size_t partials_cap = 0, sums_cap = 0;
  double *h_result = nullptr, *d_result = nullptr;  // mapped pinned host
  unsigned long long ticket_issued = 0;  // ticket of the last finalize launch (see sync_and_collect)
  int ticket_nout = 0;                   // result words that launch writes
  bool ticket_wait = true;
  // CMX_OPT_SPIN_WAIT, one policy for the three places a host thread waits (include/cmax_hip.h): spin_eval_us = how long an evaluation
  // spins on its completion ticket before it blocks in hipStreamSynchronize (-1: for as long as it takes, bounded by 20 ms);
  // spin_idle_us = how long a thread that has NOTHING on the device spins before it sleeps (a group's workers between commands, a
  // background context held behind an urgent burst)
  int spin_eval_us = -1, spin_idle_us = 50;
  size_t result_cap = 0;
  double *h_many = nullptr, *d_many = nullptr;  // cmx_*_eval_many: one 4096-double result block per evaluation of the list
  size_t many_cap = 0;
  double *result_override = nullptr;            // where the next finalize writes instead of d_result (eval_many)
  // gated gradient pass behind a cost-only evaluation (cmx_hint_next_df; cmx_pipeline.cpp run_adjoint / finish_cost_only)
  double *h_result2 = nullptr, *d_result2 = nullptr;  // second mapped result block: the gated pass reports here
  int *d_gate = nullptr;                        // written by the cost-only evaluation's finalize, read by the gated launch
  int gate_mode = 0;                            // hint for the NEXT cost-only evaluation (consumed by it): 0 none, 1..4 see gate_condition
  double gate_thr = 0;
  bool gated_df = true;                         // CMX_OPT_GATED_DF
  bool gate_arm = false;                        // set around run_adjoint(phase 3) when a gated pass will be queued behind it
  bool gated_pending = false, gated_fired = false;  // a gated pass is queued behind the last evaluation / its gate opened
  unsigned long long ticket2_issued = 0;
  int ticket2_nout = 0;
  unsigned long long gated_launches = 0, gated_hits = 0;
  // device-driven solve (cmx_chain.cpp): the FR-CG machine lives in device memory and advances inside the finalize steps
  bool chain_solve = true;            // CMX_OPT_CHAIN_SOLVE (front end)
  bool chain_self_gating = true;      // CMX_OPT_CHAIN_SOLVE 4: slots with a finalize behind the image pass and a flag-gated gradient pass (A/B)
  int chain_test = 0;                 // CMX_OPT_CHAIN_SOLVE 2 / 3: the host takes over after a few slots (exercises the hand-over)
  bool chain_active = false;          // set around the launches of a chain slot: kernels read omega / skip from device memory
  ChainDev *d_chain = nullptr;        // the machine + next evaluation point + end-of-solve flag
  double *h_chain_ring = nullptr, *d_chain_ring = nullptr;  // mapped result blocks: 2 per slot (cost stage, gradient stage)
  ChainDev *h_chain_init = nullptr;   // pinned staging of the machine's initial state (two blocks, alternating per solve)
  int chain_init_sel = 0;
  ChainDev *d_chain_init = nullptr;   // the same two blocks as the device sees them
  bool chain_warm = false;            // the last device-driven solve ended normally on self-gating slots: d_chain's flags and moment
                                      // rows, the ping-pong planes and the accumulator rows are in the state the next solve's first
                                      // slot expects -- it starts without the initial copy and without clearing anything
  bool chain_first = false;           // set while the first slot of a warm-started solve is being queued
  unsigned chain_seq = 0;             // slots executed so far on d_chain (parity = which moment rows the next slot adds to)
  double chain_x0[3] = {0, 0, 0};
  double *chain_block_a = nullptr, *chain_block_g = nullptr;  // device pointers of the blocks of the slot being queued
  int64_t chain_solves = 0, chain_slots = 0, chain_takeovers = 0, chain_warm_starts = 0;
  int tail_finalize = 1;              // CMX_OPT_TAIL_FINALIZE: 0 off, 1 on (back end: cost-only evaluations), 2 on everywhere
  bool tail_poll = false;             // ... 3: as 1 with the POLLING tail on the front-end gather (measured: no gain, profiles/r06_tail_poll.txt)
  unsigned *d_tail_counters = nullptr;  // kTailCounterWords words, all-zero between launches
  double *d_gacc = nullptr;             // kTailShards x kGaccStride gradient accumulators of the tail finalize, all-zero between launches

  // native RCCL exchange (cmx_comm_attach): every evaluation all-reduces its partial planes / gradient sums in place
  ncclComm_t comm = nullptr;
  cmx_allreduce_fn comm_fn = nullptr;  // caller-supplied transport (cmx_comm_attach_custom) instead of RCCL
  void *comm_user = nullptr;
  // internal (cmx_group.cpp's direct transport): out-of-place sum of `count` floats, in -> out, ONE host barrier; consecutive calls
  // must alternate their `in` buffer (exchange_tiles does)
  int (*comm_fn_oop)(void *user, const void *in, void *out, size_t count, void *hip_stream) = nullptr;
  // ... and its first half alone: publish `in`, ONE host barrier, wait for the peers' "send buffer complete" events on the stream, hand
  // back every member's send pointer -- the caller then sums them inside its own unpack kernel (launch_xset_sum_unpack)
  int (*comm_fn_peers)(void *user, const void *in, const void **ptrs, int *n, void *hip_stream) = nullptr;
  int comm_rank = 0, comm_size = 1;
  bool sharded() const { return comm != nullptr || comm_fn != nullptr; }
  // exchange set of the sparse plane exchange (cmx_comm.cpp): the tiles whose partial sums travel.  Two list / membership buffers
  // alternate: [xset_cur] is what this evaluation exchanges, the other one is written by xset_kernel for the next evaluation.
  int *d_xlist[2] = {nullptr, nullptr};
  unsigned char *d_xmember[2] = {nullptr, nullptr};
  int *d_xmiss = nullptr;
  size_t xset_tiles_cap = 0;
  float *d_xstage = nullptr;       // staging of the listed tiles of both planes (what the collective runs on)
  size_t xstage_cap = 0;
  // out-of-place transports (a group's one-shot direct all-reduce, RCCL): a second send buffer alternating with d_xstage (a peer may
  // still be reading the previous collective's while the next one is being packed) and the receive buffer the unpack reads
  float *d_xstage_b = nullptr, *d_xstage_out = nullptr;
  size_t xstage_b_cap = 0, xstage_out_cap = 0;
  int xstage_sel = 0;
  int xset_cur = 0, xset_n = -1;   // tiles in [xset_cur]; -1: no set known (first evaluation of a window) -> whole planes
  bool xset_pending = false;       // the last evaluation ran xset_kernel: its words wait in h_result[kXsetSlot..]
  bool xset_used = false;          // ... and exchanged the set (not the whole planes)
  unsigned long long xset_seq = 0; // sequence number of that launch (stamped into the words)
  int64_t sharded_host_syncs = 0, xset_misses = 0;
  int64_t comm_bytes_eval = 0, comm_calls_eval = 0;  // bytes / collectives of the last sharded evaluation

  // timing
  bool timing = false;
  int timing_mask = 0;  // bit i: record HIP events around kernel class i
  int timing_every = 1;           // sample every n-th evaluation
  unsigned long long timing_tick = 0;  // evaluations (accumulate calls) since timing was enabled
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  double t_ms[CMX_T_COUNT] = {0};
  int64_t t_n[CMX_T_COUNT] = {0};
};

// device-resident event store: the stream is uploaded once; packets and windows are
// cut from it on the device
struct cmx_events {
  int device = 0, W = 0, H = 0;  // device = the first replica's
  size_t capacity = 0;
  int64_t first_index = 0;   // global sequence number of slot 0
  size_t size = 0;           // events held
  // One replica of the stream per device (cmx_events_create: one; cmx_events_create_group: one per DISTINCT device of a
  // group's member list -- 40M events x 12 B is nothing in 288 GB, and every member cuts its own batch range on its own device).
  struct Replica {
    int device = 0;
    uint32_t *d_xy[2] = {nullptr, nullptr};  // x | y << 16 ; two buffers: drop_before compacts into the other one
    int64_t *d_t[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;            // uploads / compactions of the replicas run side by side
  };
  std::vector<Replica> rep;
  int cur = 0;
  uint32_t *h_xy = nullptr;  // pinned staging of a push (packed coordinates, then the timestamps): one host pass, N async uploads
  int64_t *h_tp = nullptr;
  size_t stage_cap = 0;
  std::vector<int64_t> h_t;  // host mirror of the timestamps (per-batch pose times are formed on the host)
  std::string err;
  const Replica *on(int dev) const {
    for (const Replica &r : rep)
      if (r.device == dev) return &r;
    return nullptr;
  }
};


// ---- error text + HIP call checking
int fail(cmx_ctx *c, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define HIP_TRY(c, expr)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail((c), CMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
int ensure(cmx_ctx *c, T *&ptr, size_t &cap, size_t need) {
  if (need <= cap && ptr) return CMX_OK;
  if (ptr) HIP_TRY(c, hipFree(ptr));
  ptr = nullptr;
  cap = 0;
  size_t n = need ? need : 1;
  HIP_TRY(c, hipMalloc((void **)&ptr, n * sizeof(T)));
  cap = n;
  return CMX_OK;
}

// ---- cmx_context.cpp
int bind_device(cmx_ctx *c);
int comm_probe_exchange(cmx_ctx *c, float *in, float *out, size_t count);  // cmx_comm.cpp
long long time_batch_ns(long long t_first, long long t_last);
double time_to_sec(long long t_ns);
int upload_gt1(cmx_ctx *c);
int setup_blur(cmx_ctx *c, double sigma);
hipEvent_t get_event(cmx_ctx *c);
void collect_spans(cmx_ctx *c);  // call after the stream has been synchronised
int create_common(cmx_ctx **out, int kind, int device, int W, int H, const double *lut);
// events as an array of records (cmx_aos_layout): the accessors the packing passes use when a hand-over comes from the *_aos entry points
struct EvAos {
  const unsigned char *base = nullptr;
  size_t stride = 0, ox = 0, oy = 0, os = 0, on = 0;
  inline unsigned X(int64_t i) const { uint16_t v; memcpy(&v, base + (size_t)i * stride + ox, 2); return v; }
  inline unsigned Y(int64_t i) const { uint16_t v; memcpy(&v, base + (size_t)i * stride + oy, 2); return v; }
  inline int64_t T(int64_t i) const {
    uint32_t sec, nsec;
    memcpy(&sec, base + (size_t)i * stride + os, 4);
    memcpy(&nsec, base + (size_t)i * stride + on, 4);
    return (int64_t)sec * 1000000000LL + (int64_t)nsec;
  }
  EvAos from(int64_t first) const { EvAos r = *this; r.base = base + (size_t)first * stride; return r; }
};
int make_aos(cmx_ctx *c, int64_t n, const void *events, const cmx_aos_layout *layout, EvAos *out);  // argument checks
int check_event_args(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t);
int check_events(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t, const EvAos *aos = nullptr);
int ensure_pinned_xy(cmx_ctx *c, size_t n);
int ensure_pinned_dts(cmx_ctx *c, size_t n);

// kernel_exact = true: the launcher attaches the two events to the kernel itself (hipExtLaunchKernelGGL start / stop:
// the dispatch's own begin / end timestamps, what rocprofv3 reports); otherwise the events are recorded on the
// stream around whatever the scope launches (kernel time + boundaries).
struct Span {
  cmx_ctx *c;
  TimedSpan s{};
  bool on, kernel_exact, used = false;
  Span(cmx_ctx *ctx, int cls, bool exact = false, bool enable = true)
      : c(ctx), on(enable && ctx->timing && ((ctx->timing_mask >> cls) & 1) && (ctx->timing_tick % ctx->timing_every) == 0),
        kernel_exact(exact) {
    if (on) {
      s.cls = cls;
      s.a = get_event(c);
      s.b = get_event(c);
      if (!kernel_exact) hipEventRecord(s.a, c->stream);
    }
  }
  hipEvent_t t0() { used = true; return on && kernel_exact ? s.a : nullptr; }
  hipEvent_t t1() { return on && kernel_exact ? s.b : nullptr; }
  ~Span() {
    if (!on) return;
    if (kernel_exact && !used) {  // nothing was launched with the events: give them back
      c->event_pool.push_back(s.a);
      c->event_pool.push_back(s.b);
      return;
    }
    if (!kernel_exact) hipEventRecord(s.b, c->stream);
    c->spans.push_back(s);
  }
};

// ---- cmx_pipeline.cpp
inline double *result_ptr(const cmx_ctx *c) { return c->result_override ? c->result_override : c->d_result; }
int begin_accum(cmx_ctx *c, int nplanes, size_t np, bool fast);
int ensure_accum(cmx_ctx *c, size_t need);
int do_binning(cmx_ctx *c, const FeSplatArgs *fe, const BeSplatArgs *be);
BinnedEvents binned(const cmx_ctx *c);
int ensure_fixed(cmx_ctx *c, size_t n);  // deterministic mode: n zeroed fixed-point accumulators
FeSplatArgs fe_args(const cmx_ctx *c, const double omega[3]);
BeSplatArgs be_args(const cmx_ctx *c);
bool adjoint_ok(const cmx_ctx *c);
void issue_finalize(cmx_ctx *c, FinalizeArgs &f, bool with_reduce);
bool arm_tail(cmx_ctx *c, FinalizeArgs &f, TailArgs &tail, bool gated = false);  // true: the next launch carries the finalize (no separate launch)
int attach_tiles(cmx_ctx *c, ImgArgs &a, bool may_skip);
int maybe_tile_list(cmx_ctx *c, ImgArgs &a, int reach);
int run_image_and_finalize(cmx_ctx *c, int P, float *out_blur0, float *out_blurd);
int run_adjoint(cmx_ctx *c, int P, int phase = 0);  // phase 3: image pass + cost-only finalize, Jt kept; 4: gated gradient pass
int finish_cost_only_speculative(cmx_ctx *c, int P);  // phase 3 (+ the gated pass when a hint is set), then the wait
int collect_gated(cmx_ctx *c, int P, double *contrast, double *grad, bool *served);  // a df served by the pass already queued
bool speculative_jt_ok(const cmx_ctx *c);
int sync_and_collect(cmx_ctx *c, bool ends_in_finalize = false);
bool can_reuse(const cmx_ctx *c, const double *x, int n, bool want_grad);
bool spin_for_ticket(const double *h_block, unsigned long long want, int nout, int budget_us = -1);
int fe_accumulate(cmx_ctx *c, const double omega[3], int nplanes, bool allow_fuse = false, bool allow_full = false);  // cmx_frontend.cpp
// cmx_chain.cpp: run the solve on the device as far as it goes.  `hs` = the host's machine, begun (sm_begin) with x = start;
// on return it holds the state after every evaluation the device reported.  *completed = false: the caller continues
// host-driven from hs (configuration not eligible, or the device's next point was not bitwise the host's)
int chain_run_frontend(cmx_ctx *c, FrcgSM &hs, bool *completed);
int chain_prealloc(cmx_ctx *c);  // the chain's device / mapped buffers (front-end contexts: at creation, so that no solve pays for them)
int finish_begin(cmx_ctx *c, int kind, int want_grad);
int finish_end(cmx_ctx *c, int kind, double *contrast, double *grad);
int be_ensure_time_bearings(cmx_ctx *c);  // cmx_backend.cpp: the gather's time-ordered bearing stream (once per window)
int be_first_iter(cmx_ctx *c);  // cmx_backend.cpp: IGp <- IG and alpha on the first evaluation of a window

// ---- cooperative scheduling between the contexts of this process that share a device (cmx_set_sched_class, cmx_context.cpp)
struct UrgentScope {  // around every entry point that puts an urgent context's evaluations on the device (nests)
  cmx_ctx *c;
  explicit UrgentScope(cmx_ctx *ctx);
  ~UrgentScope();
};
void yield_to_urgent(cmx_ctx *c);  // first thing of a background context's evaluation: hold it while an urgent burst is on the device

// ---- cmx_group.cpp: the entry points of the C ABI hand a group's handle to these
bool is_group(const cmx_ctx *c);
int group_size(const cmx_ctx *c);
int group_members(const cmx_ctx *c, cmx_ctx **out, int max);  // the member contexts in rank order (a plain context: itself); returns their number
int group_all(cmx_ctx *leader, const std::function<int(cmx_ctx *, int)> &fn);  // fn(member, rank) on every member; first failure
void group_destroy(cmx_ctx *leader);
int group_set_window(cmx_ctx *leader, const EvAos *aos, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns, int order, int K,
                     const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                     int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG);
int group_eval(cmx_ctx *leader, const double *drotv, double *contrast, double *grad);
int group_set_window_from(cmx_ctx *leader, const cmx_events *e, int64_t first, int64_t count, int order, int K, const double *knots_xyzw,
                          int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size,
                          int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG);
int be_eval_one(cmx_ctx *c, const double *drotv, double *contrast, double *grad);  // cmx_backend.cpp: one context's evaluation
#define CMX_NOT_FOR_GROUPS(c, what) \
  do { if ((c) && (c)->group) return fail((c), CMX_ERR_STATE, what " is not available on a group (the group runs its own exchange)"); } while (0)

// ---- cmx_comm.cpp
int finish_sharded(cmx_ctx *c, int kind, bool exchange_planes, double *contrast, double *grad);
void comm_reset_xset(cmx_ctx *c);  // a new window / packet / panorama: the next exchange covers the whole planes
void comm_release(cmx_ctx *c);  // destroys an attached communicator (cmx_destroy)

// ---- cmx_frontend.cpp / cmx_backend.cpp: the bodies behind set_packet / set_window and their *_from forms
// (d_raw != nullptr: the events are already on the device -- event store)
int fe_set_packet_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                       const uint32_t *d_raw, int64_t t_ref_ns, double fx, double fy, double cx, double cy,
                       int event_batch_size, double blur_sigma, int contrast_measure, const EvAos *aos = nullptr);
int be_set_window_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                       const uint32_t *d_raw, const int64_t *d_t, int order, int K, const double *knots_xyzw,
                       int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size,
                       int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG, const EvAos *aos = nullptr);
