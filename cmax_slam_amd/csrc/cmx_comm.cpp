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
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
int comm_allreduce(cmx_ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op = ncclSum) {
  if (!c->comm || count == 0) return CMX_OK;
  Span sp(c, CMX_T_COMM);  // on the stream: the collective itself plus the wait for the slowest rank
  const ncclResult_t r = rccl().AllReduce(buf, buf, count, dt, op, c->comm, c->stream);
  if (r != ncclSuccess) return fail(c, CMX_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
  return CMX_OK;
}

// Large panoramas: the ranks' votes cover a few tile rows of a mostly empty map.  All-reduce (max) the tile-occupancy
// flags (a few KB), read them back, and sum only the band of rows any rank touched -- 64 MB per evaluation become
// ~16 MB at 4096x2048 (BASELINE config 5).  Every rank derives the band from the same reduced flags, so the collectives
// match by construction.  Returns 1 if it handled the exchange, 0 if the caller should exchange the planes whole.
constexpr size_t kSparseExchangeMinPlaneBytes = (size_t)8 << 20;
static int exchange_touched_rows(cmx_ctx *c, int *handled) {
  *handled = 0;
  const size_t np = (size_t)c->Wp * c->Hp;
  if (c->kind != KIND_BE || !c->accum_flagged || !c->d_tflags || np * sizeof(float) < kSparseExchangeMinPlaneBytes ||
      c->accum_count != 2 * np)
    return CMX_OK;
  const int tiles_x = (c->Wp + kTileX - 1) / kTileX, tiles_y = (c->Hp + kTileY - 1) / kTileY;
  int rc = comm_allreduce(c, c->d_tflags, (size_t)tiles_x * tiles_y, ncclUint8, ncclMax);
  if (rc) return rc;
  std::vector<unsigned char> flags((size_t)tiles_x * tiles_y);
  HIP_TRY(c, hipMemcpyAsync(flags.data(), c->d_tflags, flags.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int r0 = tiles_y, r1 = -1;
  for (int ty = 0; ty < tiles_y; ty++)
    for (int tx = 0; tx < tiles_x; tx++)
      if (flags[(size_t)ty * tiles_x + tx]) { r0 = ty < r0 ? ty : r0; r1 = ty > r1 ? ty : r1; break; }
  *handled = 1;
  if (r1 < r0) return CMX_OK;  // nobody voted anywhere
  const size_t row0 = (size_t)r0 * kTileY, row1 = std::min((size_t)(r1 + 1) * kTileY, (size_t)c->Hp);
  for (int plane = 0; plane < 2; plane++) {
    rc = comm_allreduce(c, c->d_accum + plane * np + row0 * c->Wp, (row1 - row0) * c->Wp, ncclFloat);
    if (rc) return rc;
  }
  return CMX_OK;
}

}  // namespace

void comm_release(cmx_ctx *c) {
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
}

// evaluation with an attached communicator: the two exchange points of SURVEY.md section 8e, in place, on the stream
int finish_sharded(cmx_ctx *c, int kind, bool exchange_planes, double *contrast, double *grad) {
  int rc = CMX_OK;
  if (exchange_planes) {
    int handled = 0;
    rc = exchange_touched_rows(c, &handled);
    if (rc) return rc;
    if (!handled) {
      rc = comm_allreduce(c, c->d_accum, c->accum_count, ncclFloat);  // sum of the ranks' partial planes
      c->accum_flagged = false;  // the planes now hold other ranks' votes this rank's occupancy flags know nothing about
    }
    if (rc) return rc;
  }
  rc = finish_begin(c, kind, grad != nullptr);
  if (rc) return rc;
  if (c->pending_P > 0) {
    rc = comm_allreduce(c, c->d_gsum, (size_t)2 * c->pending_P, ncclDouble);  // adjoint mode: S1,S2 partial sums
    if (rc) return rc;
  }
  return finish_end(c, kind, contrast, grad);
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
  c->comm_size = 1;
  c->comm_rank = 0;
  return CMX_OK;
}

