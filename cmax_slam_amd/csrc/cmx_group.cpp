// cmx_group.cpp -- one-process multi-GPU: a GROUP of back-end contexts behind ONE handle, ONE host thread, ONE optimiser.
//
// The reference's host is one process with one back-end thread and one GSL instance
// (src/cmax_slam.cpp:92, src/backend/global_optim_contrast_gsl.cpp:23-33): it cannot be launched once per GPU.  A group is the
// form that drops into it: cmx_backend_create_group() returns an ordinary cmx_ctx handle; the unchanged global_contrast_fdf
// body calls cmx_backend_eval(handle, ...) and the evaluation fans out to the N member contexts -- member r holds the
// contiguous range of whole event batches dist.batch_range gives rank r (event_pano_warper.cpp:188-196 is the loop being
// sharded), the members exchange their partial planes after the splat exactly as the one-process-per-GPU form does (cmx_comm.cpp:
// same collective, same exchange set); the gradient rows need no collective here -- the gradient is linear in them, every member
// finalizes its own and the calling thread adds the members' gradients -- and ONE contrast / gradient comes back.  No replicated optimisers, no launcher.
//
// How the fan-out runs.  Queueing one member's evaluation is ~10 launches + 2 collectives of host work (~40-60 us with RCCL's
// enqueue path); eight members queued by one thread would take longer than the ~150 us the evaluation runs for.  So every
// member but the first has a WORKER thread owned by the group, parked on the group's command word; the calling thread
// publishes the command, runs member 0 itself, and collects.  Between the calls of a solve the workers spin (hand-over
// ~0.2 us); after 50 us without a command (CMX_OPT_SPIN_WAIT) they sleep on a condition variable: an idle group holds no core.
// The caller sees a synchronous, single-threaded API.
//
// Transports (what an all-reduce between the members is):
//   CMX_GROUP_RCCL    ncclCommInitAll over the members' devices, one communicator per member, used from its worker
//                     (RCCL's "one thread per device" mode) -- the default whenever the members sit on different devices;
//   CMX_GROUP_DIRECT  peer-to-peer kernels, no library.  The production exchange (a staged tile set below 2 MB) is ONE-SHOT: every
//                     member packs into one of two alternating send buffers, records one event, meets its peers at one host
//                     barrier, and its unpack kernel sums ALL members' send buffers (16-byte loads, member order: the same bits
//                     everywhere) straight into its own planes (direct_peers + cmx_kernels.hip: xset_sum_unpack_kernel).  Whole
//                     planes (a window's first evaluation, small panoramas) and the u8 / f64 collectives keep reduce-scatter +
//                     all-gather in place (every slice summed by exactly one member), ordered by HIP events between the members'
//                     streams.  Members on ONE device (what a single-GPU box can run: the tests, bench.py's group leg,
//                     tools/soak_group.py) always use it; across devices it needs peer access (xGMI), opens every peer-reading
//                     kernel with a system-scope acquire, and has never run on hardware -- opt-in.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only (see cmx_comm.cpp)

#include "cmx_context.hpp"

namespace {

constexpr int kMaxMembers = 16;
constexpr double kBarrierTimeoutMs = 20000.0;
constexpr double kGroupCallTimeoutMs = 120000.0;  // the calling thread's bound on ONE fanned-out call (group_all)

// ---- the direct transport's kernels.  ptrs[m] = member m's buffer (same length everywhere); member `me` owns slice `me`.
// xdev: the members sit on more than one device -- every kernel that reads a peer's buffer then opens with a SYSTEM-scope acquire (its
// own L2 may hold lines of the peer's buffer from the previous collective; the producer's side is the event's release to system scope)
struct PeerPtrs { void *p[kMaxMembers]; int xdev; };
__device__ __forceinline__ void peer_acquire(int xdev) {
  if (xdev) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Every access is ONE 16-byte load / store per lane (global_load_dwordx4): across xGMI a 4-byte access per lane is the worst width
// there is.  Slices start on 16-byte boundaries (direct_launch); the last (count % (16 / sizeof T)) elements go one by one.
template <typename T>
struct alignas(16) Vec16 { T v[16 / sizeof(T)]; };
template <typename T, bool MAX>
__device__ inline T red(T a, T b) { return MAX ? (b > a ? b : a) : (T)(a + b); }

template <typename T, bool MAX>
__global__ __launch_bounds__(256) void peer_reduce_scatter_kernel(PeerPtrs pp, int n, int me, size_t beg, size_t end) {
  peer_acquire(pp.xdev);
  constexpr int L = 16 / sizeof(T);
  T *mine = static_cast<T *>(pp.p[me]);
  const size_t nv = (end - beg) / L;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    const size_t at = beg + i * L;
    Vec16<T> acc = *reinterpret_cast<const Vec16<T> *>(static_cast<const T *>(pp.p[0]) + at);
    for (int m = 1; m < n; m++) {  // member order: whoever computes a slice, the bits are the same
      const Vec16<T> v = *reinterpret_cast<const Vec16<T> *>(static_cast<const T *>(pp.p[m]) + at);
#pragma unroll
      for (int k = 0; k < L; k++) acc.v[k] = red<T, MAX>(acc.v[k], v.v[k]);
    }
    *reinterpret_cast<Vec16<T> *>(mine + at) = acc;
  }
  for (size_t i = beg + nv * L + (size_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (size_t)gridDim.x * 256) {
    T acc = static_cast<const T *>(pp.p[0])[i];
    for (int m = 1; m < n; m++) acc = red<T, MAX>(acc, static_cast<const T *>(pp.p[m])[i]);
    mine[i] = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void peer_all_gather_kernel(PeerPtrs pp, int n, int me, size_t per, size_t count) {
  peer_acquire(pp.xdev);
  constexpr int L = 16 / sizeof(T);
  T *mine = static_cast<T *>(pp.p[me]);
  for (int m = 0; m < n; m++) {
    if (m == me) continue;
    const size_t beg = (size_t)m * per < count ? (size_t)m * per : count, end = beg + per < count ? beg + per : count;
    const T *src = static_cast<const T *>(pp.p[m]);
    const size_t nv = (end - beg) / L;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256)
      *reinterpret_cast<Vec16<T> *>(mine + beg + i * L) = *reinterpret_cast<const Vec16<T> *>(src + beg + i * L);
    for (size_t i = beg + nv * L + (size_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (size_t)gridDim.x * 256) mine[i] = src[i];
  }
}
// ONE-SHOT all-reduce of a staged message (every production exchange: the tile set, < 2 MB): every member reads ALL members' send
// buffers over the whole range and writes the sum, added in member order (the same bits everywhere), to its OWN receive buffer.
// Nobody writes a buffer a peer reads: one host barrier, one set of event waits, no gather phase (cmx_group.cpp: direct_oneshot).
__global__ __launch_bounds__(256) void peer_sum_oneshot_kernel(PeerPtrs in, float *__restrict out, int n, size_t count) {
  peer_acquire(in.xdev);
  const size_t nv = count / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    float4 acc = static_cast<const float4 *>(in.p[0])[i];
    for (int m = 1; m < n; m++) {
      const float4 v = static_cast<const float4 *>(in.p[m])[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4 *>(out)[i] = acc;
  }
  for (size_t i = nv * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    float acc = static_cast<const float *>(in.p[0])[i];
    for (int m = 1; m < n; m++) acc += static_cast<const float *>(in.p[m])[i];
    out[i] = acc;
  }
}

// ---- RCCL entry points a group needs beyond cmx_comm.cpp's (same lazy loading)
struct RcclGroupApi {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclGroupApi &rccl_group() {
  static RcclGroupApi api = [] {
    RcclGroupApi a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) return a;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.handle, "ncclCommInitAll");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.ok = a.CommInitAll && a.GetErrorString;
    return a;
  }();
  return api;
}

double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct cmx_group {
  int n = 0;
  int transport = CMX_GROUP_DIRECT;
  cmx_ctx *m[kMaxMembers] = {nullptr};
  bool cross_device = false;  // members on more than one device (the peer kernels' acquire scope)
  std::vector<std::thread> workers;
  // ---- command word: the leader publishes, the workers run it on their member
  std::atomic<unsigned long long> seq{0};
  const std::function<int(cmx_ctx *, int)> *cmd = nullptr;
  std::atomic<int> remaining{0};
  int rc[kMaxMembers] = {0};
  std::atomic<int> abort_flag{0};  // a member failed: peers waiting for it in the direct transport give up
  bool quit = false;
  std::string setup_err;
  bool ready = false;  // the workers are running: until then (and after a failed set-up) no call may fan out -- nothing would collect it
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> sleepers{0};
  // ---- direct transport
  struct DirectUser { cmx_group *g; int rank; } duser[kMaxMembers];
  void *slot_ptr[kMaxMembers] = {nullptr};
  hipEvent_t ev_ready[kMaxMembers] = {nullptr}, ev_rs[kMaxMembers] = {nullptr}, ev_ag[kMaxMembers] = {nullptr};
  // one-shot (out-of-place) collectives: send pointers and "my send buffer is complete" events, two sets alternating per collective
  // (a member may publish collective k+1 while a slower peer is still launching collective k)
  const void *os_in[2][kMaxMembers] = {{nullptr}};
  hipEvent_t ev_os[2][kMaxMembers] = {{nullptr}};
  unsigned long long os_seq[kMaxMembers] = {0};
  std::atomic<int> bar_count{0};
  std::atomic<unsigned> bar_gen{0};
  // ---- results of the members of the call in flight
  double out_c[kMaxMembers] = {0};
  std::vector<double> out_g[kMaxMembers];
  int64_t evals = 0;
  double last_fanout_us = 0;  // host time of the last fan-out: command published -> every member returned
  // ---- CMX_GROUP_AUTO: both transports set up where both are possible, the faster one (measured on this group's own message) kept
  ncclComm_t rccl_comms[kMaxMembers] = {nullptr};
  bool have_rccl = false, have_direct = false, calibrated = false;
  double calib_us[3] = {-1.0, -1.0, -1.0};  // per exchange of kCalibFloats floats, indexed by CMX_GROUP_* (-1: not available / not measured)
};

namespace {

// sense-reversing spin barrier over the n member threads of a call; false: a peer failed (abort) or timed out
bool group_barrier(cmx_group *g) {
  const unsigned gen = g->bar_gen.load(std::memory_order_acquire);
  if (g->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == g->n) {
    g->bar_count.store(0, std::memory_order_relaxed);
    g->bar_gen.store(gen + 1, std::memory_order_release);
    return true;
  }
  const double t0 = now_us();
  for (unsigned spins = 0;; spins++) {
    if (g->bar_gen.load(std::memory_order_acquire) != gen) return true;
    if (g->abort_flag.load(std::memory_order_relaxed)) return false;
    __builtin_ia32_pause();
    if ((spins & 4095u) == 4095u && now_us() - t0 > kBarrierTimeoutMs * 1e3) return false;
  }
}

template <typename T>
void direct_launch(cmx_group *g, int me, size_t count, int op, hipStream_t s, bool gather_phase) {
  PeerPtrs pp{};
  pp.xdev = g->cross_device ? 1 : 0;
  for (int k = 0; k < g->n; k++) pp.p[k] = g->slot_ptr[k];
  // slices of whole 16-byte groups; member r owns [r*per, (r+1)*per)
  const size_t grp = 16 / sizeof(T);
  const size_t per = ((count + g->n - 1) / g->n + grp - 1) / grp * grp;
  const size_t beg = (size_t)me * per < count ? (size_t)me * per : count, end = beg + per < count ? beg + per : count;
  if (!gather_phase) {
    if (end <= beg) return;
    int blocks = (int)((end - beg + 255) / 256);
    blocks = blocks > 1024 ? 1024 : blocks;
    if (op == CMX_OP_MAX) hipLaunchKernelGGL((peer_reduce_scatter_kernel<T, true>), dim3(blocks), dim3(256), 0, s, pp, g->n, me, beg, end);
    else hipLaunchKernelGGL((peer_reduce_scatter_kernel<T, false>), dim3(blocks), dim3(256), 0, s, pp, g->n, me, beg, end);
  } else {
    int blocks = (int)((per + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL((peer_all_gather_kernel<T>), dim3(blocks), dim3(256), 0, s, pp, g->n, me, per, count);
  }
}

// cmx_allreduce_fn of the direct transport: called by member `rank`'s thread with the member's device current
int direct_allreduce(void *user, void *buf, size_t count, int dt, int op, void *hip_stream) {
  auto *u = static_cast<cmx_group::DirectUser *>(user);
  cmx_group *g = u->g;
  const int me = u->rank, n = g->n;
  hipStream_t s = (hipStream_t)hip_stream;
  g->slot_ptr[me] = buf;
  if (hipEventRecord(g->ev_ready[me], s) != hipSuccess) return 1;  // my partial sums are complete behind this point
  if (!group_barrier(g)) return 2;                                  // every pointer published, every ready event recorded
  for (int k = 0; k < n; k++)
    if (k != me && hipStreamWaitEvent(s, g->ev_ready[k], 0) != hipSuccess) return 1;
  if (dt == CMX_DT_U8) direct_launch<unsigned char>(g, me, count, op, s, false);
  else if (dt == CMX_DT_F32) direct_launch<float>(g, me, count, op, s, false);
  else direct_launch<double>(g, me, count, op, s, false);
  if (hipEventRecord(g->ev_rs[me], s) != hipSuccess) return 1;
  if (!group_barrier(g)) return 2;
  for (int k = 0; k < n; k++)
    if (k != me && hipStreamWaitEvent(s, g->ev_rs[k], 0) != hipSuccess) return 1;  // every slice holds its final sum
  if (dt == CMX_DT_U8) direct_launch<unsigned char>(g, me, count, op, s, true);
  else if (dt == CMX_DT_F32) direct_launch<float>(g, me, count, op, s, true);
  else direct_launch<double>(g, me, count, op, s, true);
  if (hipEventRecord(g->ev_ag[me], s) != hipSuccess) return 1;
  if (!group_barrier(g)) return 2;
  // nothing queued after the collective may overwrite my buffer while a peer is still copying my slice out of it
  for (int k = 0; k < n; k++)
    if (k != me && hipStreamWaitEvent(s, g->ev_ag[k], 0) != hipSuccess) return 1;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// The one-shot form (cmx_ctx::comm_fn_oop; called by member `rank`'s thread, the member's device current).  Per collective: ONE host
// barrier and n-1 stream waits.  Why that is enough: (1) a member reads the peers' SEND buffers and writes only its own RECEIVE buffer, so
// no peer ever waits for this member's kernel; (2) the caller alternates two send buffers, and a member's "ready" event of collective
// k+1 is recorded on its stream behind its kernel of collective k -- whoever has waited for the peers' events of collective k+1 knows
// that every read of its collective-k send buffer is over before it packs collective k+2 into the same buffer.
// first half: publish, barrier, waits; ptrs[k] = member k's send buffer (cmx_ctx::comm_fn_peers)
int direct_peers(void *user, const void *in, const void **ptrs, int *n_out, void *hip_stream) {  // *n_out: members, | 0x100 when they span devices
  auto *u = static_cast<cmx_group::DirectUser *>(user);
  cmx_group *g = u->g;
  const int me = u->rank, n = g->n;
  hipStream_t s = (hipStream_t)hip_stream;
  const int par = (int)(g->os_seq[me]++ & 1ull);
  g->os_in[par][me] = in;
  if (hipEventRecord(g->ev_os[par][me], s) != hipSuccess) return 1;  // my send buffer is complete behind this point
  if (!group_barrier(g)) return 2;                                   // every pointer published, every event recorded
  for (int k = 0; k < n; k++) {
    ptrs[k] = g->os_in[par][k];
    if (k != me && hipStreamWaitEvent(s, g->ev_os[par][k], 0) != hipSuccess) return 1;
  }
  *n_out = n | (g->cross_device ? 0x100 : 0);
  return 0;
}
int direct_oneshot(void *user, const void *in, void *out, size_t count, void *hip_stream) {
  const void *ptrs[kMaxMembers];
  int n = 0;
  const int rc = direct_peers(user, in, ptrs, &n, hip_stream);
  if (rc) return rc;
  n &= 0xff;
  hipStream_t s = (hipStream_t)hip_stream;
  PeerPtrs pp{};
  pp.xdev = static_cast<cmx_group::DirectUser *>(user)->g->cross_device ? 1 : 0;
  for (int k = 0; k < n; k++) pp.p[k] = const_cast<void *>(ptrs[k]);
  int blocks = (int)((count / 4 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(peer_sum_oneshot_kernel, dim3(blocks), dim3(256), 0, s, pp, static_cast<float *>(out), n, count);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

void worker_loop(cmx_group *g, int rank) {
  unsigned long long seen = 0;
  for (;;) {
    // wait for the next command: spin first (the calls of a solve follow one another within microseconds), then sleep
    // (CMX_OPT_SPIN_WAIT: 50 us by default -- the evaluations of a solve follow one another within a few microseconds, and an idle
    //  group must not hold a core per member; 0: straight to sleep)
    const double t0 = now_us(), idle_us = (double)g->m[rank]->spin_idle_us;
    bool have = g->seq.load(std::memory_order_acquire) != seen;
    for (unsigned spins = 0; !have && idle_us > 0; spins++) {
      if (g->seq.load(std::memory_order_acquire) != seen) { have = true; break; }
      __builtin_ia32_pause();
      if ((spins & 63u) == 63u && now_us() - t0 > idle_us) break;
    }
    if (!have) {
      std::unique_lock<std::mutex> lk(g->mu);
      g->sleepers.fetch_add(1, std::memory_order_seq_cst);
      g->cv.wait(lk, [&] { return g->seq.load(std::memory_order_seq_cst) != seen; });
      g->sleepers.fetch_sub(1, std::memory_order_seq_cst);
    }
    seen = g->seq.load(std::memory_order_acquire);
    if (g->quit) return;
    int rc = CMX_ERR_STATE;
    if (g->cmd) rc = (*g->cmd)(g->m[rank], rank);
    g->rc[rank] = rc;
    if (rc) g->abort_flag.store(1, std::memory_order_relaxed);
    g->remaining.fetch_sub(1, std::memory_order_acq_rel);
  }
}

void publish(cmx_group *g) {
  g->seq.fetch_add(1, std::memory_order_seq_cst);
  if (g->sleepers.load(std::memory_order_seq_cst) > 0) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->cv.notify_all();
  }
}

void batch_range(int64_t n, int B, int rank, int world, int64_t *beg, int64_t *end) {  // cmax_slam_amd/dist.py: batch_range
  const int64_t nb = (n + B - 1) / B, per = (nb + world - 1) / world;
  const int64_t b0 = std::min<int64_t>((int64_t)rank * per, nb), b1 = std::min<int64_t>(b0 + per, nb);
  *beg = std::min<int64_t>(b0 * B, n);
  *end = std::min<int64_t>(b1 * B, n);
}

}  // namespace

bool is_group(const cmx_ctx *c) { return c && c->group && c->group_rank == 0; }
int group_size(const cmx_ctx *c) { return (c && c->group) ? c->group->n : 1; }
int group_members(const cmx_ctx *c, cmx_ctx **out, int max) {
  if (!c) return 0;
  if (!c->group) { if (max > 0) out[0] = const_cast<cmx_ctx *>(c); return 1; }
  const int n = c->group->n < max ? c->group->n : max;
  for (int r = 0; r < n; r++) out[r] = c->group->m[r];
  return n;
}

// run fn(member, rank) on every member -- member 0 on the calling thread, the others on their workers -- and return the first
// failure (its text copied to the handle)
int group_all(cmx_ctx *leader, const std::function<int(cmx_ctx *, int)> &fn) {
  cmx_group *g = leader->group;
  if (!g->ready)
    return fail(leader, CMX_ERR_STATE, "the group was not set up (cmx_backend_create_group failed: %s): only cmx_destroy is valid on this handle",
                g->setup_err.c_str());
  const double t0 = now_us();
  g->abort_flag.store(0, std::memory_order_relaxed);
  g->bar_count.store(0, std::memory_order_relaxed);  // (a call that failed half-way may have left arrivals behind)
  g->cmd = &fn;
  g->remaining.store(g->n - 1, std::memory_order_release);
  if (g->n > 1) publish(g);
  int rc0 = fn(g->m[0], 0);
  g->rc[0] = rc0;
  if (rc0) g->abort_flag.store(1, std::memory_order_relaxed);
  // the workers' calls are bounded (a member waits at most kBarrierTimeoutMs in a barrier, a handful of barriers per call): the caller
  // is too.  A member that has not returned by then is a hung device -- the handle refuses further calls (only cmx_destroy stays valid).
  const double wait0 = now_us();
  for (unsigned spins = 0; g->remaining.load(std::memory_order_acquire) > 0; spins++) {
    __builtin_ia32_pause();
    if ((spins & 4095u) == 4095u && now_us() - wait0 > kGroupCallTimeoutMs * 1e3) {
      g->abort_flag.store(1, std::memory_order_relaxed);
      g->ready = false;
      g->setup_err = "a member did not return from a call within " + std::to_string((int)(kGroupCallTimeoutMs / 1000)) + " s";
      return fail(leader, CMX_ERR_HIP, "group call timed out: %d member(s) still running after %.0f s; the handle is unusable",
                  g->remaining.load(std::memory_order_acquire), kGroupCallTimeoutMs / 1000);
    }
  }
  g->cmd = nullptr;
  g->last_fanout_us = now_us() - t0;
  int first_bad = -1;
  for (int r = 0; r < g->n && first_bad < 0; r++)
    if (g->rc[r]) first_bad = r;
  if (first_bad < 0) return CMX_OK;
  // A failed call may have left the members out of step: the one-shot exchange takes its buffer / event parity from per-member
  // counters, and a member that failed before its exchange did not advance its own while its peers did.  Every member is
  // back (nothing of theirs runs host-side); drain their streams and restart the parities from zero.
  for (int r = 0; r < g->n; r++) {
    if (hipSetDevice(g->m[r]->device) == hipSuccess && g->m[r]->stream) (void)hipStreamSynchronize(g->m[r]->stream);
    g->os_seq[r] = 0;
  }
  (void)hipSetDevice(g->m[0]->device);
  const int r = first_bad;
  if (r != 0) leader->err = "member " + std::to_string(r) + " (device " + std::to_string(g->m[r]->device) + "): " + g->m[r]->err;
  return g->rc[r];
}

static void group_teardown(cmx_group *g) {
  if (!g) return;
  if (!g->workers.empty()) {
    g->quit = true;
    publish(g);
    for (auto &t : g->workers) t.join();
    g->workers.clear();
  }
  for (int r = 0; r < g->n; r++) {
    if (g->m[r]) (void)hipSetDevice(g->m[r]->device);
    if (g->ev_ready[r]) hipEventDestroy(g->ev_ready[r]);
    if (g->ev_rs[r]) hipEventDestroy(g->ev_rs[r]);
    if (g->ev_ag[r]) hipEventDestroy(g->ev_ag[r]);
    for (int k = 0; k < 2; k++)
      if (g->ev_os[k][r]) hipEventDestroy(g->ev_os[k][r]);
  }
  for (int r = g->n - 1; r >= 0; r--) {  // members: the ordinary single-context destroy (communicators included)
    cmx_ctx *m = g->m[r];
    if (!m) continue;
    m->group = nullptr;
    if (m->comm_fn) { m->comm_fn = nullptr; m->comm_fn_oop = nullptr; m->comm_fn_peers = nullptr; m->comm_user = nullptr; }
    if (g->rccl_comms[r]) m->comm = g->rccl_comms[r];  // (set up but not selected: the member's destroy releases it all the same)
    cmx_destroy(m);
  }
  delete g;
}
void group_destroy(cmx_ctx *leader) { group_teardown(leader->group); }

// ---- transport set-up.  CMX_GROUP_RCCL / CMX_GROUP_DIRECT set up the one asked for; CMX_GROUP_AUTO sets up every transport the
// devices allow and keeps the one that moves THIS group's message fastest (group_calibrate).
static int setup_rccl(cmx_group *g, cmx_ctx *leader, const int *devices, int n_devices) {
  if (!rccl_group().ok) return fail(leader, CMX_ERR_HIP, "librccl.so.1 could not be loaded (ncclCommInitAll)");
  const ncclResult_t r = rccl_group().CommInitAll(g->rccl_comms, n_devices, devices);
  if (r != ncclSuccess) return fail(leader, CMX_ERR_HIP, "ncclCommInitAll failed: %s", rccl_group().GetErrorString(r));
  g->have_rccl = true;
  return CMX_OK;
}
static int setup_direct(cmx_group *g, cmx_ctx *leader, const int *devices, int n_devices) {
  for (int a = 0; a < n_devices; a++) {  // peer access between the members' devices (a no-op on one device)
    HIP_TRY(leader, hipSetDevice(devices[a]));
    for (int b = 0; b < n_devices; b++) {
      if (devices[a] == devices[b]) continue;
      int can = 0;
      HIP_TRY(leader, hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
      if (!can) return fail(leader, CMX_ERR_HIP, "device %d cannot access device %d: the direct transport needs peer access", devices[a], devices[b]);
      const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(leader, CMX_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d) failed", devices[a], devices[b]);
      (void)hipGetLastError();
    }
  }
  for (int r = 0; r < n_devices; r++) {
    HIP_TRY(leader, hipSetDevice(devices[r]));
    // system-scope release at the record: what a peer DEVICE reads behind the wait must be in memory, not in this device's L2
    const unsigned flags = hipEventDisableTiming | hipEventReleaseToSystem;
    HIP_TRY(leader, hipEventCreateWithFlags(&g->ev_ready[r], flags));
    HIP_TRY(leader, hipEventCreateWithFlags(&g->ev_rs[r], flags));
    HIP_TRY(leader, hipEventCreateWithFlags(&g->ev_ag[r], flags));
    HIP_TRY(leader, hipEventCreateWithFlags(&g->ev_os[0][r], flags));
    HIP_TRY(leader, hipEventCreateWithFlags(&g->ev_os[1][r], flags));
    g->duser[r].g = g;
    g->duser[r].rank = r;
  }
  g->have_direct = true;
  return CMX_OK;
}
// point every member at one of the transports that have been set up
static void select_transport(cmx_group *g, int transport) {
  for (int r = 0; r < g->n; r++) {
    cmx_ctx *m = g->m[r];
    m->comm_rank = r;
    m->comm_size = g->n;
    if (transport == CMX_GROUP_RCCL) {
      m->comm = g->rccl_comms[r];
      m->comm_fn = nullptr; m->comm_fn_oop = nullptr; m->comm_fn_peers = nullptr; m->comm_user = nullptr;
    } else {
      m->comm = nullptr;
      m->comm_fn = direct_allreduce; m->comm_fn_oop = direct_oneshot; m->comm_fn_peers = direct_peers; m->comm_user = &g->duser[r];
    }
  }
  g->transport = transport;
}
// Time the staged exchange of a message the size of a production tile set (config 4: ~1 MB) through every transport that is set up:
// kCalibWarm + kCalibReps exchanges each, every member on its own thread exactly as an evaluation issues them, wall clock of the
// calling thread around the fan-out (stream-synchronised at the end).  A transport that fails its probe is dropped.
constexpr size_t kCalibFloats = 256 * 1024;
constexpr int kCalibWarm = 3, kCalibReps = 20;
static int group_calibrate(cmx_group *g, cmx_ctx *leader) {
  float *buf[kMaxMembers][3] = {{nullptr}};
  int rc = group_all(leader, [&](cmx_ctx *m, int r) {
    int rc2 = bind_device(m);
    for (int k = 0; k < 3 && !rc2; k++) {
      if (hipMalloc((void **)&buf[r][k], kCalibFloats * sizeof(float)) != hipSuccess) rc2 = fail(m, CMX_ERR_HIP, "calibration buffer");
      else if (hipMemsetAsync(buf[r][k], 0, kCalibFloats * sizeof(float), m->stream) != hipSuccess) rc2 = fail(m, CMX_ERR_HIP, "calibration buffer");
    }
    return rc2;
  });
  for (int t = CMX_GROUP_RCCL; t <= CMX_GROUP_DIRECT && !rc; t++) {
    if ((t == CMX_GROUP_RCCL && !g->have_rccl) || (t == CMX_GROUP_DIRECT && !g->have_direct)) continue;
    select_transport(g, t);
    for (int pass = 0; pass < 2; pass++) {
      const int reps = pass == 0 ? kCalibWarm : kCalibReps;
      const double t0 = now_us();
      const int prc = group_all(leader, [&](cmx_ctx *m, int r) {
        int rc2 = bind_device(m);
        for (int k = 0; k < reps && !rc2; k++) rc2 = comm_probe_exchange(m, buf[r][k & 1], buf[r][2], kCalibFloats);
        if (!rc2 && hipStreamSynchronize(m->stream) != hipSuccess) rc2 = fail(m, CMX_ERR_HIP, "calibration exchange");
        return rc2;
      });
      if (prc) { g->calib_us[t] = -1.0; if (t == CMX_GROUP_RCCL) g->have_rccl = false; else g->have_direct = false; break; }
      if (pass == 1) g->calib_us[t] = (now_us() - t0) / reps;
    }
  }
  (void)group_all(leader, [&](cmx_ctx *m, int r) {
    (void)bind_device(m);
    for (int k = 0; k < 3; k++) if (buf[r][k]) (void)hipFree(buf[r][k]);
    return CMX_OK;
  });
  if (rc) return rc;
  if (!g->have_rccl && !g->have_direct) return fail(leader, CMX_ERR_HIP, "no transport passed its calibration exchange");
  int best = g->have_direct ? CMX_GROUP_DIRECT : CMX_GROUP_RCCL;
  if (g->have_rccl && g->have_direct && g->calib_us[CMX_GROUP_RCCL] < g->calib_us[CMX_GROUP_DIRECT]) best = CMX_GROUP_RCCL;
  select_transport(g, best);
  g->calibrated = true;
  for (int r = 0; r < g->n; r++) { g->m[r]->comm_bytes_eval = 0; g->m[r]->comm_calls_eval = 0; }
  return CMX_OK;
}

// transport set-up + worker threads of a group whose members exist; on failure the group stays "not ready" (group_all refuses)
static int group_connect(cmx_group *g, cmx_ctx *leader, const int *devices, int n_devices, int transport, bool same_device) {
  const bool automatic = transport == CMX_GROUP_AUTO;
  int rc = CMX_OK;
  if (transport == CMX_GROUP_RCCL) {
    if (same_device) return fail(leader, CMX_ERR_INVALID_ARG, "RCCL cannot place two ranks on one device: use CMX_GROUP_DIRECT");
    rc = setup_rccl(g, leader, devices, n_devices);
  } else if (transport == CMX_GROUP_DIRECT) {
    rc = setup_direct(g, leader, devices, n_devices);
  } else {  // AUTO: whatever the devices allow (members sharing a device: the direct transport only); a transport that cannot be
            // set up (no peer access, no librccl) is simply not a candidate
    const int rd = setup_direct(g, leader, devices, n_devices);
    const int rr = same_device ? CMX_ERR_INVALID_ARG : setup_rccl(g, leader, devices, n_devices);
    if (rd && rr) rc = rd;  // (the text of the direct transport's failure; RCCL's is in the handle's error string only if it came last)
    else leader->err.clear();
  }
  if (rc) return rc;
  select_transport(g, g->have_direct && (transport == CMX_GROUP_DIRECT || automatic) ? CMX_GROUP_DIRECT : CMX_GROUP_RCCL);
  for (int r = 1; r < n_devices; r++) g->workers.emplace_back(worker_loop, g, r);
  g->ready = true;
  if (automatic) {
    rc = group_calibrate(g, leader);
    if (rc) { g->ready = false; return rc; }
  }
  HIP_TRY(leader, hipSetDevice(devices[0]));
  return CMX_OK;
}

static std::atomic<int> g_diag_force_cross_device{0};
int cmx_diag_set(int key, int value) {
  if (key != CMX_DIAG_FORCE_CROSS_DEVICE) return CMX_ERR_INVALID_ARG;
  g_diag_force_cross_device.store(value != 0 ? 1 : 0, std::memory_order_relaxed);
  return CMX_OK;
}

int cmx_backend_create_group(cmx_ctx **out, const int *devices, int n_devices, int W, int H, const double *lut, int Wp, int Hp,
                             int transport) {
  if (!out) return CMX_ERR_INVALID_ARG;
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > kMaxMembers || transport < CMX_GROUP_AUTO || transport > CMX_GROUP_DIRECT)
    return CMX_ERR_INVALID_ARG;
  if (n_devices == 1) return cmx_backend_create(out, devices[0], W, H, lut, Wp, Hp);  // a group of one IS a plain context
  bool same_device = false;
  for (int a = 0; a < n_devices; a++)
    for (int b = a + 1; b < n_devices; b++) same_device = same_device || devices[a] == devices[b];
  cmx_group *g = new cmx_group();
  g->n = n_devices;
  g->transport = transport == CMX_GROUP_AUTO ? CMX_GROUP_DIRECT : transport;  // (AUTO: replaced by the measured choice in group_connect)
  for (int a = 1; a < n_devices; a++) g->cross_device = g->cross_device || devices[a] != devices[0];
  if (g_diag_force_cross_device.load(std::memory_order_relaxed)) g->cross_device = true;  // (cmax_hip_diag.h: a one-GPU box runs the xdev paths)
  int rc = CMX_OK;
  for (int r = 0; r < n_devices && !rc; r++) {
    rc = cmx_backend_create(&g->m[r], devices[r], W, H, lut, Wp, Hp);
    if (g->m[r]) { g->m[r]->group = g; g->m[r]->group_rank = r; }
  }
  cmx_ctx *leader = g->m[0];
  *out = leader;  // returned on failure too when it exists: the caller reads cmx_last_error and destroys it
  if (rc) {
    if (!leader) { group_teardown(g); return rc; }
    for (int r = 1; r < n_devices; r++)
      if (g->m[r] && !g->m[r]->err.empty()) { leader->err = g->m[r]->err; break; }
    g->setup_err = leader->err;
    return rc;
  }
  rc = group_connect(g, leader, devices, n_devices, transport, same_device);
  if (rc) g->setup_err = leader->err;  // the handle is returned (the caller reads cmx_last_error and destroys it); every other call on it fails with CMX_ERR_STATE
  return rc;
}

// ---- the group forms of the entry points (called from the C ABI functions when the handle is a group's)
int group_set_window(cmx_ctx *leader, const EvAos *aos, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns, int order, int K,
                     const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                     int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG) {
  cmx_group *g = leader->group;
  const int N = g->n;
  const bool shardable = n > 0 && event_batch_size > 0 && ((x && y && t_ns) || aos);
  const int rc = group_all(leader, [&](cmx_ctx *m, int r) {
    int64_t beg = 0, end = (r == 0) ? n : 0;  // bad arguments: member 0 gets them as they are and reports the error
    if (shardable) {
      batch_range(n, event_batch_size, r, N, &beg, &end);
      // One event more than the member's batches hold.  The reference's loop `for (beg = 0; beg < n - 1; beg += B)` with
      // `end = (n - beg > B) ? beg + B : n` (event_pano_warper.cpp:188-196) never opens a batch for a single trailing event;
      // with the extra event a member's last batch is a whole one (n - beg = B + 1 > B) and the event itself is in no batch of
      // this member -- the next member owns it.  The last member holding events sees the true tail, quirk included.
      if (end > beg && end < n) end += 1;
    }
    const bool none = !shardable && r != 0;
    if (aos) {  // the member's range of the host's records (cmx_backend_set_window_aos)
      const EvAos mine = aos->from(none ? 0 : beg);
      return be_set_window_impl(m, none ? 0 : end - beg, nullptr, nullptr, nullptr, nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns,
                                num_fixed, t_next_win_beg_ns, event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG, &mine);
    }
    return be_set_window_impl(m, none ? 0 : end - beg, none ? nullptr : x + beg, none ? nullptr : y + beg, none ? nullptr : t_ns + beg,
                              nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size,
                              event_sample_rate, blur_sigma, contrast_measure, IG);
  });
  if (rc)  // a window one member rejected is no window: no member may walk into a collective its peers will not join
    for (int r = 0; r < N; r++) g->m[r]->have_data = false;
  return rc;
}

int group_member_window_from(cmx_ctx *m, const cmx_events *e, int64_t first, int64_t beg, int64_t end, int order, int K,
                             const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                             int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG);  // cmx_events.cpp

int group_set_window_from(cmx_ctx *leader, const cmx_events *e, int64_t first, int64_t count, int order, int K, const double *knots_xyzw,
                          int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size,
                          int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG) {
  cmx_group *g = leader->group;
  const int N = g->n;
  if (!e) return fail(leader, CMX_ERR_INVALID_ARG, "null event store");
  for (int r = 0; r < N; r++)
    if (!e->on(g->m[r]->device))
      return fail(leader, CMX_ERR_INVALID_ARG, "the event store holds no replica on device %d (member %d): create it with cmx_events_create_group",
                  g->m[r]->device, r);
  const bool shardable = count > 0 && event_batch_size > 0;
  const int rc = group_all(leader, [&](cmx_ctx *m, int r) {
    int64_t beg = 0, end = (r == 0) ? count : 0;
    if (shardable) {
      batch_range(count, event_batch_size, r, N, &beg, &end);
      if (end > beg && end < count) end += 1;  // (group_set_window: the member's last batch is a whole one, the extra event is in none of its batches)
    }
    return group_member_window_from(m, e, first, beg, end, order, K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns,
                                    event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG);
  });
  if (rc)
    for (int r = 0; r < N; r++) g->m[r]->have_data = false;
  return rc;
}

int group_eval(cmx_ctx *leader, const double *drotv, double *contrast, double *grad) {
  cmx_group *g = leader->group;
  const int P = 3 * (leader->K - leader->num_fixed);
  for (int r = 0; r < g->n; r++)
    if ((int)g->out_g[r].size() < P + 1) g->out_g[r].assign((size_t)P + 1, 0.0);
  const int rc = group_all(leader, [&](cmx_ctx *m, int r) { return be_eval_one(m, drotv, &g->out_c[r], grad ? g->out_g[r].data() : nullptr); });
  if (rc) return rc;
  g->evals++;
  // every member finished on the same all-reduced planes: the contrast is the same bits everywhere (cmx_comm.cpp) -- a difference
  // means the members' states diverged and nothing they report can be trusted.  The gradient rows were NOT exchanged
  // (finish_exchanged): member r reports (2/N)(S1_r - mu S2_r) over its own events, and the whole gradient is their sum, added
  // here in member order (the same bits in every run that splits the window the same way).
  for (int r = 1; r < g->n; r++) {
    if (memcmp(&g->out_c[r], &g->out_c[0], sizeof(double)) != 0)
      return fail(leader, CMX_ERR_STATE, "group members disagree (member %d: contrast %.17g vs %.17g)", r, g->out_c[r], g->out_c[0]);
  }
  if (contrast) *contrast = g->out_c[0];
  const bool partial = g->m[0]->group_partial_grad;  // (rank-invariant: a function of the options)
  if (grad)
    for (int k = 0; k < P; k++) {
      double s = g->out_g[0][k];
      for (int r = 1; r < g->n && partial; r++) s += g->out_g[r][k];
      grad[k] = s;
    }
  return CMX_OK;
}

int cmx_group_transport_info(cmx_ctx *c, int *chosen, int *measured, double *us_direct, double *us_rccl) {
  if (!c) return CMX_ERR_INVALID_ARG;
  const cmx_group *g = c->group;
  if (chosen) *chosen = g ? g->transport : CMX_GROUP_AUTO;
  if (measured) *measured = (g && g->calibrated) ? 1 : 0;
  if (us_direct) *us_direct = g ? g->calib_us[CMX_GROUP_DIRECT] : -1.0;
  if (us_rccl) *us_rccl = g ? g->calib_us[CMX_GROUP_RCCL] : -1.0;
  return CMX_OK;
}
int cmx_group_info(cmx_ctx *c, int *n_members, int *devices, int max_devices, int *transport, int64_t *events_per_member,
                   double *last_fanout_us) {
  if (!c) return CMX_ERR_INVALID_ARG;
  const cmx_group *g = c->group;
  const int n = g ? g->n : 1;
  if (n_members) *n_members = n;
  if (transport) *transport = g ? g->transport : CMX_GROUP_AUTO;
  if (last_fanout_us) *last_fanout_us = g ? g->last_fanout_us : 0.0;
  for (int r = 0; r < n && r < max_devices; r++) {
    const cmx_ctx *m = g ? g->m[r] : c;
    if (devices) devices[r] = m->device;
    if (events_per_member) events_per_member[r] = m->n_packed;
  }
  return CMX_OK;
}
