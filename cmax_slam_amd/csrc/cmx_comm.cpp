// cmx_comm.cpp -- the RCCL communicator inside the evaluator: sharded evaluations exchange their partial planes and
// partial gradient sums in place, on the context's stream.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is dlopen()ed when a communicator is first attached

#include "cmx_context.hpp"

namespace {

// ---- RCCL, loaded lazily so that single-GPU hosts carry no dependency on it
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi &rccl() {
  static RcclApi api = [] {
    RcclApi a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");  // optional: fuses the per-plane band collectives
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
int comm_allreduce(cmx_ctx *c, void *buf, size_t count, int dt /* CMX_DT_* */, int op = CMX_OP_SUM) {
  if (!c->sharded() || count == 0) return CMX_OK;
  c->comm_bytes_eval += (int64_t)count * (dt == CMX_DT_U8 ? 1 : (dt == CMX_DT_F32 ? 4 : 8));
  c->comm_calls_eval++;
  Span sp(c, CMX_T_COMM);  // on the stream: the collective itself plus the wait for the slowest rank
  if (c->comm_fn) {
    const int r = c->comm_fn(c->comm_user, buf, count, dt, op, (void *)c->stream);
    if (r != 0) return fail(c, CMX_ERR_HIP, "caller-supplied all-reduce failed with status %d", r);
    return CMX_OK;
  }
  const ncclDataType_t ndt = dt == CMX_DT_U8 ? ncclUint8 : (dt == CMX_DT_F32 ? ncclFloat : ncclDouble);
  const ncclResult_t r = rccl().AllReduce(buf, buf, count, ndt, op == CMX_OP_MAX ? ncclMax : ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) return fail(c, CMX_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
  return CMX_OK;
}

// Exchange of the partial planes between splat and blur.
// Which collectives are issued depends on RANK-INVARIANT state only -- context kind, plane size, what the accumulate call
// produced (a function of the options and of the call itself), the row band every rank derived from the same all-reduced
// flags -- never on a rank's own event count: mismatched collectives are undefined behaviour in RCCL.
//
// Large panoramas (planes of 8 MB and more): the ranks' votes cover a few tile rows of a mostly empty map.  The
// tile-occupancy flags (a few KB) are all-reduced with max; band_kernel reduces them to the first / last touched tile
// row and writes that to mapped host memory; the planes are then summed over the band of rows the PREVIOUS evaluation
// found, widened by kBandMargin tile rows (the whole plane while no band is known) -- 64 MB per evaluation become ~16 MB
// at 4096x2048 (BASELINE config 5) with no host synchronisation between splat and blur.  band_kernel also reports whether
// a touched row lay outside the band that was exchanged; finish_sharded() then completes the evaluation with a whole-plane exchange (rare: the
// parameters moved the votes by more than two tile rows between two evaluations).
constexpr size_t kSparseExchangeMinPlaneBytes = (size_t)8 << 20;
constexpr int kBandMargin = 2;
static int allreduce_rows(cmx_ctx *c, int tile_row0, int tile_row1 /* exclusive */) {
  const size_t np = (size_t)c->Wp * c->Hp;
  const size_t row0 = (size_t)tile_row0 * kTileY, row1 = std::min((size_t)tile_row1 * kTileY, (size_t)c->Hp);
  if (row1 <= row0) return CMX_OK;
  // the two planes' bands are two regions of memory: one RCCL group = one launch instead of two (native communicator only;
  // a caller-supplied transport sees the two calls)
  const bool group = c->comm && rccl().GroupStart && rccl().GroupEnd;
  if (group) rccl().GroupStart();
  int rc = CMX_OK;
  for (int plane = 0; plane < 2 && !rc; plane++)
    rc = comm_allreduce(c, c->d_accum + plane * np + row0 * c->Wp, (row1 - row0) * c->Wp, CMX_DT_F32);
  if (group && rccl().GroupEnd() != ncclSuccess && !rc) rc = fail(c, CMX_ERR_HIP, "ncclGroupEnd failed");
  return rc;
}
static int exchange_planes(cmx_ctx *c) {
  const size_t np = (size_t)c->Wp * c->Hp;
  c->band_pending = false;
  const bool sparse = c->kind == KIND_BE && c->accum_flagged && c->d_tflags && np * sizeof(float) >= kSparseExchangeMinPlaneBytes &&
                      c->accum_count == 2 * np;
  if (!sparse) {
    int rc = comm_allreduce(c, c->d_accum, c->accum_count, CMX_DT_F32);  // sum of the ranks' partial planes
    // the planes now hold other ranks' votes this rank's occupancy flags know nothing about: rebuild the flags from the
    // summed planes (one small launch) so that the image passes keep skipping the empty tiles of the panorama -- without
    // them the image pass of BASELINE config 4 (1024^2) took 26.6 us instead of 10.6
    if (!rc && c->kind == KIND_BE && c->accum_flagged && c->d_tflags && c->accum_count == 2 * np) {
      launch_tile_flags_pair(c->d_accum, c->d_accum + np, c->Wp, c->Hp, c->d_tflags, c->stream);
      HIP_TRY(c, hipGetLastError());
    } else {
      c->accum_flagged = false;
    }
    return rc;
  }
  const int tiles_x = (c->Wp + kTileX - 1) / kTileX, tiles_y = (c->Hp + kTileY - 1) / kTileY;
  int rc = comm_allreduce(c, c->d_tflags, (size_t)tiles_x * tiles_y, CMX_DT_U8, CMX_OP_MAX);
  if (rc) return rc;
  int lo = c->band_lo, hi = c->band_hi;
  if (hi < lo) { lo = 0; hi = tiles_y - 1; }  // no band known yet (first evaluation of a window): the whole plane
  launch_band(c->d_tflags, tiles_x, tiles_y, lo, hi, c->d_result + kBandSlot, ++c->band_seq, c->stream);
  HIP_TRY(c, hipGetLastError());
  rc = allreduce_rows(c, lo, hi + 1);
  if (rc) return rc;
  c->band_pending = true;
  c->band_used_lo = lo;
  c->band_used_hi = hi;
  return CMX_OK;
}

}  // namespace

void comm_release(cmx_ctx *c) {
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_fn = nullptr;
}
void comm_reset_band(cmx_ctx *c) {
  c->band_lo = 0;
  c->band_hi = -1;
  c->band_pending = false;
}

static int finish_exchanged(cmx_ctx *c, int kind, double *contrast, double *grad) {
  // the gradient sums travel as the accumulator rows the kernels add to (kTailShards x kGaccStride doubles, 25 KB: the
  // collective is latency-bound either way) whenever run_adjoint can use them; otherwise as the 2P-double buffer
  c->shard_acc = true;
  int rc = finish_begin(c, kind, grad != nullptr);
  if (!rc && c->pending_P > 0) {
    const bool rows = !c->deterministic && c->d_gacc && 2 * c->pending_P <= kGaccStride;  // (run_adjoint's condition: rank-invariant)
    rc = rows ? comm_allreduce(c, c->d_gacc, (size_t)kTailShards * kGaccStride, CMX_DT_F64)
              : comm_allreduce(c, c->d_gsum, (size_t)2 * c->pending_P, CMX_DT_F64);  // adjoint mode: S1,S2 partial sums
  }
  if (!rc) rc = finish_end(c, kind, contrast, grad);
  c->shard_acc = false;
  return rc;
}

// evaluation with an attached communicator: the two exchange points of SURVEY.md section 8e, in place, on the stream
int finish_sharded(cmx_ctx *c, int kind, bool exchange, double *contrast, double *grad) {
  int rc = CMX_OK;
  c->comm_bytes_eval = 0;  // what THIS evaluation exchanges (cmx_get_stats [8], [9]... see the header)
  c->comm_calls_eval = 0;
  c->gate_mode = 0;  // (no gated pass with a communicator attached: a hint must not outlive this evaluation)
  c->gated_pending = false;
  if (exchange) {
    rc = exchange_planes(c);
    if (rc) return rc;
  }
  rc = finish_exchanged(c, kind, contrast, grad);
  if (rc || !exchange || !c->band_pending) return rc;
  // the results are on the host, and with them what band_kernel found (written by an earlier kernel of the same stream)
  c->band_pending = false;
  const int tiles_y = (c->Hp + kTileY - 1) / kTileY;
  // ... in an EARLIER kernel than the finalize whose ticket was waited for, outside its checksummed snapshot: the words carry
  // their own stamp.  A rank that read a stale band would choose another exchange than its peers (a hang), so a snapshot
  // that does not verify is re-read, then the stream is synchronised, and only then is it an error.
  double bw[3] = {0, 0, 0};
  bool band_ok = false;
  for (int attempt = 0; attempt < 3 && !band_ok; attempt++) {
    for (int spin = 0; spin < 20000 && !band_ok; spin++) {
      const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(c->h_result + kBandSlot);
      const unsigned long long b0 = w[0], b1 = w[1], b2 = w[2], st = w[3];
      if ((b0 ^ b1 ^ b2 ^ (c->band_seq * kTicketMix)) == st) {
        memcpy(&bw[0], &b0, 8); memcpy(&bw[1], &b1, 8); memcpy(&bw[2], &b2, 8);
        band_ok = true;
      }
    }
    if (!band_ok) HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (!band_ok) return fail(c, CMX_ERR_HIP, "the band kernel's result did not arrive (sequence %llu)", c->band_seq);
  const int r0 = (int)bw[0], r1 = (int)bw[1];
  const bool miss = bw[2] != 0.0;
  if (r1 >= r0) {
    c->band_lo = std::max(0, r0 - kBandMargin);
    c->band_hi = std::min(tiles_y - 1, r1 + kBandMargin);
  } else {
    comm_reset_band(c);  // nobody voted anywhere
  }
  if (miss) {
    // every rank read the same three numbers (they derive from the all-reduced flags), so every rank is here: the rows
    // outside the exchanged band still hold partial sums -- exchange them and finish once more on the complete planes
    c->band_misses++;
    rc = allreduce_rows(c, 0, c->band_used_lo);
    if (rc) return rc;
    rc = allreduce_rows(c, c->band_used_hi + 1, tiles_y);
    if (rc) return rc;
    rc = finish_exchanged(c, kind, contrast, grad);
  }
  return rc;
}

// ---- native RCCL communicator (one process per GPU; the launcher distributes the 128-byte id)
int cmx_comm_unique_id(char id[CMX_COMM_ID_BYTES]) {
  if (!id) return CMX_ERR_INVALID_ARG;
  if (!rccl().ok) return CMX_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) <= CMX_COMM_ID_BYTES, "id buffer too small");
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return CMX_ERR_HIP;
  memset(id, 0, CMX_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return CMX_OK;
}
int cmx_comm_attach(cmx_ctx *c, const char id[CMX_COMM_ID_BYTES], int rank, int nranks) {
  if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, CMX_ERR_INVALID_ARG, "bad communicator arguments");
  if (!rccl().ok) return fail(c, CMX_ERR_HIP, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
  int rc = bind_device(c);
  if (rc) return rc;
  if (c->comm) { rccl().CommDestroy(c->comm); c->comm = nullptr; }
  c->comm_fn = nullptr;
  comm_reset_band(c);
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return fail(c, CMX_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); }
  c->comm_rank = rank;
  c->comm_size = nranks;
  return CMX_OK;
}
int cmx_comm_detach(cmx_ctx *c) {
  if (!c) return CMX_ERR_INVALID_ARG;
  int rc = bind_device(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_fn = nullptr;
  c->comm_user = nullptr;
  c->comm_size = 1;
  c->comm_rank = 0;
  comm_reset_band(c);
  return CMX_OK;
}
int cmx_comm_attach_custom(cmx_ctx *c, cmx_allreduce_fn fn, void *user, int rank, int nranks) {
  if (!c || !fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, CMX_ERR_INVALID_ARG, "bad communicator arguments");
  int rc = cmx_comm_detach(c);
  if (rc) return rc;
  c->comm_fn = fn;
  c->comm_user = user;
  c->comm_rank = rank;
  c->comm_size = nranks;
  return CMX_OK;
}

