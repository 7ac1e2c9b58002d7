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
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
    a.CommCount = (decltype(a.CommCount))dlsym(a.handle, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.handle, "ncclCommUserRank");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
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
// produced (a function of the options and of the call itself), the size of the exchange set every rank derived from the same
// all-reduced flags -- never on a rank's own event count: mismatched collectives are undefined behaviour in RCCL.
//
// Panoramas (planes of 1 MB and more): a window's votes cover a few per cent of the map (BASELINE config 4: ~3 % of the tiles of
// two 4 MB planes; config 5: ~1 % of two 32 MB planes), and every rank's slab of the window covers a part of that.  So the
// tile-occupancy flags (a few KB) are all-reduced with max first, and what travels is the EXCHANGE SET: the tiles any rank
// flagged in the PREVIOUS evaluation, dilated (xset_kernel) -- packed from both planes into one staging buffer with the map behind
// them (as floats: summed, > 0 = occupied), summed by ONE collective, unpacked.  Its size is known on the host from the previous evaluation's result words, so nothing is waited for
// between splat and blur.  xset_kernel also lists the flagged tiles the set did NOT cover (the parameters moved the votes
// further than the dilation); finish_sharded() then completes the evaluation with a second exchange of exactly those tiles.
// The first evaluation of a window, and sets of more than half of the map, exchange the whole planes.
constexpr size_t kSparseExchangeMinPlaneBytes = (size_t)1 << 20;
static int ensure_xset(cmx_ctx *c, size_t ntiles) {
  if (ntiles <= c->xset_tiles_cap && c->d_xmiss) return CMX_OK;
  void **ptrs[5] = {(void **)&c->d_xlist[0], (void **)&c->d_xlist[1], (void **)&c->d_xmember[0], (void **)&c->d_xmember[1],
                    (void **)&c->d_xmiss};
  const size_t bytes[5] = {ntiles * sizeof(int), ntiles * sizeof(int), ntiles, ntiles, ntiles * sizeof(int)};
  for (int k = 0; k < 5; k++) {
    if (*ptrs[k]) HIP_TRY(c, hipFree(*ptrs[k]));
    *ptrs[k] = nullptr;
    HIP_TRY(c, hipMalloc(ptrs[k], bytes[k]));
  }
  c->xset_tiles_cap = ntiles;
  c->xset_n = -1;  // (whatever set was known lived in the old buffers)
  return CMX_OK;
}
// Sum of `count` floats across the ranks, staged: `in` is this rank's packed contribution, *result is where the sum is afterwards.
// RCCL and a group's direct transport are out of place (in -> out: nobody writes a buffer a peer may be reading, so the direct
// transport needs ONE host barrier and one set of event waits per collective, cmx_group.cpp); a caller-supplied all-reduce is in place.
static int comm_allreduce_staged(cmx_ctx *c, float *in, float *out, size_t count, float **result) {
  *result = in;
  if (!c->sharded() || count == 0) return CMX_OK;
  if (c->comm_fn && !c->comm_fn_oop) return comm_allreduce(c, in, count, CMX_DT_F32);
  c->comm_bytes_eval += (int64_t)count * 4;
  c->comm_calls_eval++;
  Span sp(c, CMX_T_COMM);
  *result = out;
  if (c->comm_fn_oop) {
    const int r = c->comm_fn_oop(c->comm_user, in, out, count, (void *)c->stream);
    if (r != 0) return fail(c, CMX_ERR_HIP, "the group's one-shot all-reduce failed with status %d", r);
    return CMX_OK;
  }
  const ncclResult_t r = rccl().AllReduce(in, out, count, ncclFloat, ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) return fail(c, CMX_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
  return CMX_OK;
}
// both planes' tiles of `list` (and, with `flags`, the occupancy map as floats behind them) -> staging -> ONE all-reduce -> back
static int exchange_tiles(cmx_ctx *c, const int *list, int n, unsigned char *flags, int ntiles) {
  if (n <= 0 && !flags) return CMX_OK;
  const size_t np = (size_t)c->Wp * c->Hp;
  const size_t need = (size_t)2 * (n > 0 ? n : 0) * kTileX * kTileY + (flags ? (size_t)ntiles : 0);
  const bool oop = c->comm_fn_oop || (c->comm && !c->comm_fn);
  // out-of-place transports: the buffers are sized ONCE for the largest set this panorama can produce (every tile of both planes + the
  // map) -- a peer of the one-shot transport may still be reading the previous collective's send buffer, which must never be freed under it
  // (`need` counts whole padded 64 x 16 tiles -- with Wp % 64 or Hp % 16 != 0 a large set exceeds 2 * Wp * Hp; the true
  //  maximum is every tile of both planes, padded, plus the map.  An out-of-place buffer is never regrown: a list that exceeds the
  //  map's own tile count is a caller error, not a reason to free memory a peer may be reading.)
  const size_t tiles_all = (size_t)((c->Wp + kTileX - 1) / kTileX) * ((c->Hp + kTileY - 1) / kTileY);
  const size_t cap = oop ? 2 * tiles_all * kTileX * kTileY + tiles_all : need;
  if (oop && need > cap) return fail(c, CMX_ERR_STATE, "exchange set of %d tiles exceeds the panorama's %zu", n, tiles_all);
  int rc = ensure(c, c->d_xstage, c->xstage_cap, cap);
  if (!rc && oop) rc = ensure(c, c->d_xstage_b, c->xstage_b_cap, cap);
  if (!rc && oop && !c->comm_fn_peers) rc = ensure(c, c->d_xstage_out, c->xstage_out_cap, cap);
  if (rc) return rc;
  float *in = c->d_xstage;
  if (oop) {
    c->xstage_sel ^= 1;
    in = c->xstage_sel ? c->d_xstage_b : c->d_xstage;
  }
  launch_xset_copy(false, c->d_accum, np, c->Wp, c->Hp, list, n > 0 ? n : 0, in, flags, ntiles, c->stream);
  if (c->comm_fn_peers) {  // a group's direct transport: the sum over the members happens inside the unpack kernel
    c->comm_bytes_eval += (int64_t)need * 4;
    c->comm_calls_eval++;
    Span sp(c, CMX_T_COMM);
    const void *ptrs[16];
    XsetPeers peers{};
    const int r = c->comm_fn_peers(c->comm_user, in, ptrs, &peers.n, (void *)c->stream);
    if (r != 0) return fail(c, CMX_ERR_HIP, "the group's one-shot exchange failed with status %d", r);
    peers.xdev = (peers.n >> 8) & 1;
    peers.n &= 0xff;
    for (int m = 0; m < peers.n; m++) peers.p[m] = static_cast<const float *>(ptrs[m]);
    launch_xset_sum_unpack(peers, c->d_accum, np, c->Wp, c->Hp, list, n > 0 ? n : 0, flags, ntiles, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  float *sum = in;
  rc = comm_allreduce_staged(c, in, c->d_xstage_out, need, &sum);
  if (rc) return rc;
  launch_xset_copy(true, c->d_accum, np, c->Wp, c->Hp, list, n > 0 ? n : 0, sum, flags, ntiles, c->stream);
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}
static int exchange_planes(cmx_ctx *c) {
  const size_t np = (size_t)c->Wp * c->Hp;
  c->xset_pending = false;
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
  const int ntiles = tiles_x * tiles_y;
  int rc = ensure_xset(c, (size_t)ntiles);
  if (rc) return rc;
  const int cur = c->xset_cur, n = c->xset_n;
  const bool use_set = n >= 0 && 2 * n <= ntiles;  // (rank-invariant: n derives from the all-reduced flags of the previous evaluation)
  if (use_set) {  // the set does not depend on THIS evaluation's flags: map and tiles travel as one collective
    rc = exchange_tiles(c, c->d_xlist[cur], n, c->d_tflags, ntiles);
  } else {
    rc = comm_allreduce(c, c->d_tflags, (size_t)ntiles, CMX_DT_U8, CMX_OP_MAX);
    if (!rc) rc = comm_allreduce(c, c->d_accum, c->accum_count, CMX_DT_F32);
  }
  if (rc) return rc;
  launch_xset(c->d_tflags, tiles_x, tiles_y, use_set ? c->d_xmember[cur] : nullptr, c->d_xlist[cur ^ 1], c->d_xmember[cur ^ 1], c->d_xmiss,
              c->d_result + kXsetSlot, ++c->xset_seq, c->stream);
  HIP_TRY(c, hipGetLastError());
  c->xset_pending = true;
  c->xset_used = use_set;
  return CMX_OK;
}

}  // namespace

// One staged exchange of `count` floats through whatever transport the context has attached -- what a group's transport calibration
// times (cmx_group.cpp group_calibrate); the caller alternates `in` between two buffers like exchange_tiles does.
int comm_probe_exchange(cmx_ctx *c, float *in, float *out, size_t count) {
  float *res = nullptr;
  return comm_allreduce_staged(c, in, out, count, &res);
}

void comm_release(cmx_ctx *c) {
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_fn = nullptr;
}
void comm_reset_xset(cmx_ctx *c) {
  c->xset_n = -1;
  c->xset_pending = false;
}

static int finish_exchanged(cmx_ctx *c, int kind, double *contrast, double *grad) {
  // the gradient sums travel as the accumulator rows the kernels add to (kTailShards x kGaccStride doubles, 25 KB: the
  // collective is latency-bound either way) whenever run_adjoint can use them; otherwise as the 2P-double buffer
  c->shard_acc = true;
  int rc = finish_begin(c, kind, grad != nullptr);
  // a GROUP's members skip this collective: the gradient is linear in the rows ((2/N)(S1 - mu S2), mu from the all-reduced
  // planes: the same on every member), so each member finalizes its own rows and the group's one host thread adds the members'
  // gradients in member order (cmx_group.cpp: group_eval) -- one collective per evaluation instead of two
  c->group_partial_grad = c->group != nullptr && c->pending_P > 0;  // (derivative planes: all-reduced with the IWE -- every member's gradient is the whole one)
  if (!rc && c->pending_P > 0 && !c->group) {
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
  if (rc || !exchange || !c->xset_pending) return rc;
  // the results are on the host, and with them what xset_kernel found (written by an earlier kernel of the same stream)
  c->xset_pending = false;
  // ... in an EARLIER kernel than the finalize whose ticket was waited for, outside its checksummed snapshot: the words carry
  // their own stamp.  A rank that read stale words would choose another exchange than its peers (a hang), so a snapshot
  // that does not verify is re-read, then the stream is synchronised, and only then is it an error.
  double xw[3] = {0, 0, 0};
  bool words_ok = false;
  for (int attempt = 0; attempt < 3 && !words_ok; attempt++) {
    for (int spin = 0; spin < 20000 && !words_ok; spin++) {
      const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(c->h_result + kXsetSlot);
      const unsigned long long b0 = w[0], b1 = w[1], b2 = w[2], st = w[3];
      if ((b0 ^ b1 ^ b2 ^ (c->xset_seq * kTicketMix)) == st) {
        memcpy(&xw[0], &b0, 8); memcpy(&xw[1], &b1, 8); memcpy(&xw[2], &b2, 8);
        words_ok = true;
      }
    }
    if (!words_ok) HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (!words_ok) {
    // This rank cannot know which exchange its peers chose next: the communicator is UNUSABLE after this error (the peers
    // would hang in their next collective) -- the caller must tear the job down.  The set is dropped so that a retry on a fresh
    // communicator starts from whole planes like every other rank.
    comm_reset_xset(c);
    return fail(c, CMX_ERR_HIP, "the exchange-set kernel's result did not arrive (sequence %llu): communicator unusable", c->xset_seq);
  }
  const int n_next = (int)xw[0], n_miss = (int)xw[1];
  c->xset_cur ^= 1;  // what xset_kernel wrote is the next evaluation's set
  c->xset_n = n_next;
  if (c->xset_used && n_miss > 0) {
    // every rank read the same numbers (they derive from the all-reduced flags), so every rank is here: the listed tiles still
    // hold partial sums -- exchange exactly those and finish once more on the complete planes
    c->xset_misses++;
    rc = exchange_tiles(c, c->d_xmiss, n_miss, nullptr, 0);
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
  CMX_NOT_FOR_GROUPS(c, "attaching a communicator");
  if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, CMX_ERR_INVALID_ARG, "bad communicator arguments");
  if (!rccl().ok) return fail(c, CMX_ERR_HIP, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
  int rc = bind_device(c);
  if (rc) return rc;
  if (c->comm) { rccl().CommDestroy(c->comm); c->comm = nullptr; }
  c->comm_fn = nullptr;
  comm_reset_xset(c);
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
  CMX_NOT_FOR_GROUPS(c, "detaching the communicator");
  int rc = bind_device(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_fn = nullptr;
  c->comm_user = nullptr;
  c->comm_size = 1;
  c->comm_rank = 0;
  comm_reset_xset(c);
  return CMX_OK;
}
int cmx_comm_info(cmx_ctx *c, int *rank, int *nranks, int *transport) {
  if (!c) return CMX_ERR_INVALID_ARG;
  int r = 0, n = 1, t = 0;
  if (c->comm) {
    t = 1;
    r = c->comm_rank;
    n = c->comm_size;
    // the live communicator's own view (a mismatch with what the launcher believes is exactly what a caller wants to see)
    if (rccl().CommCount && rccl().CommCount(c->comm, &n) != ncclSuccess) return fail(c, CMX_ERR_HIP, "ncclCommCount failed");
    if (rccl().CommUserRank && rccl().CommUserRank(c->comm, &r) != ncclSuccess) return fail(c, CMX_ERR_HIP, "ncclCommUserRank failed");
  } else if (c->comm_fn) {
    t = 2;
    r = c->comm_rank;
    n = c->comm_size;
  }
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  if (transport) *transport = t;
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

