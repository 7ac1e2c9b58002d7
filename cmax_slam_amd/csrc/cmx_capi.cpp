// cmx_capi.cpp -- the extern "C" boundary declared in include/cmax_hip.h: context management, event upload
// (SoA packing, per-batch time tables), evaluation sequencing on a HIP stream.  All compute is in
// cmx_kernels.hip; there is NO CPU fallback: every entry point fails loudly if HIP is unavailable.
#include "../../include/cmax_hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is dlopen()ed when a communicator is first attached

#include "cmx_internal.hpp"

using namespace cmx;

namespace {

enum { KIND_FE = 1, KIND_BE = 2 };
constexpr long long kMaxEvents = 1LL << 30;  // kernels index events with 32-bit ints (grid-stride loops add up to 2^19)

struct TimedSpan { int cls; hipEvent_t a, b; };

}  // namespace

struct cmx_ctx {
  int kind = 0, device = 0;
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
  double *d_gpartials = nullptr;
  size_t gpartials_cap = 0;
  double *d_vparts = nullptr;  // back end: per-batch partial V sums of the gather pass
  size_t vparts_cap = 0;
  double *d_gsum = nullptr;   // this rank's partial gradient sums [P] (caller-owned when external: RCCL reduces it in place)
  size_t gsum_cap = 0;
  bool gsum_external = false;
  bool finish_pending = false; // finish_begin ran, finish_end has not
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
  bool x_valid = false;       // plane 0 (and the pose table) hold the accumulation for last_x
  int reuse_image = 1;        // df right after f at the same point reuses the image (CMX_OPT_REUSE_IMAGE)
  int64_t reuse_hits = 0;
  double last_x[3 * kMaxKnots] = {0};  // parameters of the last accumulate (the gather pass re-warps the events)

  // LDS-privatised splat (CMX_OPT_SPLAT_MODE = 1): events sorted by destination tile, chunk table
  bool bin_valid = false;
  uint32_t *d_keys = nullptr, *d_keys_s = nullptr, *d_idx = nullptr, *d_idx_s = nullptr, *d_sxy = nullptr, *d_sbatch = nullptr;
  size_t bin_cap = 0;
  void *d_sort_temp = nullptr;
  size_t sort_temp_cap = 0;
  int *d_tile_start = nullptr;
  size_t tile_start_cap = 0;
  Chunk *d_chunks = nullptr;
  size_t chunks_cap = 0;
  int nchunks = 0;          // launch bound of the chunk table (its true length lives in d_nchunks)
  int *d_nchunks = nullptr;
  bool nchunks_exact = false;  // nchunks has been replaced by the table's true length (read back after the first evaluation)
  unsigned *d_fallback = nullptr;
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
  size_t result_cap = 0;

  // native RCCL exchange (cmx_comm_attach): every evaluation all-reduces its partial planes / gradient sums in place
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 1;

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
  int device = 0, W = 0, H = 0;
  size_t capacity = 0;
  int64_t first_index = 0;   // global sequence number of slot 0
  size_t size = 0;           // events held
  uint32_t *d_xy[2] = {nullptr, nullptr};  // x | y << 16 ; two buffers: drop_before compacts into the other one
  int64_t *d_t[2] = {nullptr, nullptr};
  int cur = 0;
  std::vector<int64_t> h_t;  // host mirror of the timestamps (per-batch pose times are formed on the host)
  std::string err;
};

namespace {

int fail(cmx_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    c->err = buf;
  }
  return code;
}

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

int bind(cmx_ctx *c) {
  HIP_TRY(c, hipSetDevice(c->device));
  return CMX_OK;
}

// ---- ros::Time arithmetic (roscpp noetic semantics), needed to reproduce the per-batch pose time:
//   time_batch = time_first + (time_last - time_first) * 0.5         [Duration*double -> fromSec: floor + round]
//   reference: local_image_warped_events.cpp:68-75, event_pano_warper.cpp:239-242
long long time_batch_ns(long long t_first, long long t_last) {
  const long long d = t_last - t_first;
  long long ds = d / 1000000000LL, dn = d % 1000000000LL;
  if (dn < 0) { dn += 1000000000LL; ds -= 1; }
  const double half = ((double)ds + 1e-9 * (double)dn) * 0.5;
  const long long hs = (long long)floor(half);
  const long long hn = (long long)round((half - (double)hs) * 1e9);
  return t_first + hs * 1000000000LL + hn;
}
double time_to_sec(long long t_ns) {  // ros::Time::toSec
  return (double)(t_ns / 1000000000LL) + 1e-9 * (double)(t_ns % 1000000000LL);
}

// c1d[q] = (G^T 1)_q for one axis of length L: taps that stay inside + the taps the forward pass reflected back
// (see adjoint_kernel / image_adjoint_kernel).  1 in the interior; only the outer r pixels differ.
int upload_gt1(cmx_ctx *c) {
  const int r = c->radius;
  for (int axis = 0; axis < 2; axis++) {
    const int L = axis == 0 ? c->imgW : c->imgH;
    if (L <= 0) continue;
    std::vector<float> v((size_t)L);
    for (int q = 0; q < L; q++) {
      double s = 0;
      for (int j = -r; j <= r; j++)
        if (q - j >= 0 && q - j < L) s += (double)c->taps[r + j];
      if (L > 2 * r + 1) {
        if (1 <= q && q <= r)
          for (int m = 0; m <= r - q; m++) s += (double)c->taps[r + q + m];
        if (L - 1 - r <= q && q <= L - 2) {
          const int d = L - 1 - q;
          for (int m = 0; m <= r - d; m++) s += (double)c->taps[r + d + m];
        }
      }
      v[(size_t)q] = (float)s;
    }
    float *&dst = axis == 0 ? c->d_cx : c->d_cy;
    size_t &cap = axis == 0 ? c->cx_cap : c->cy_cap;
    int rc = ensure(c, dst, cap, (size_t)L);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(dst, v.data(), (size_t)L * sizeof(float), hipMemcpyHostToDevice));
  }
  return CMX_OK;
}

// cv::GaussianBlur(Size(0,0), sigma) on CV_32F: ksize = cvRound(sigma*8+1)|1; fp64 kernel normalised, cast to fp32
int setup_blur(cmx_ctx *c, double sigma) {
  c->sigma = sigma;
  if (!(sigma > 0)) {
    c->radius = 0;
    c->taps[0] = 1.f;
    return upload_gt1(c);
  }
  const int n = ((int)lrint(sigma * 4 * 2 + 1)) | 1;
  const int r = n / 2;
  if (r > kMaxRadius) return fail(c, CMX_ERR_INVALID_ARG, "blur_sigma %.3f needs radius %d > %d", sigma, r, kMaxRadius);
  double t[2 * kMaxRadius + 1], sum = 0;
  const double scale2X = -0.5 / (sigma * sigma);
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    t[i] = exp(scale2X * x * x);
    sum += t[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) c->taps[i] = (float)(t[i] * sum);
  c->radius = r;
  return upload_gt1(c);
}

// ---- timing helpers
hipEvent_t get_event(cmx_ctx *c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}
// kernel_exact = true: the launcher attaches the two events to the kernel itself (hipExtLaunchKernelGGL start / stop:
// the dispatch's own begin / end timestamps, what rocprofv3 reports); otherwise the events are recorded on the
// stream around whatever the scope launches (kernel time + boundaries).
struct Span {
  cmx_ctx *c;
  TimedSpan s{};
  bool on, kernel_exact, used = false;
  Span(cmx_ctx *ctx, int cls, bool exact = false)
      : c(ctx), on(ctx->timing && ((ctx->timing_mask >> cls) & 1) && (ctx->timing_tick % ctx->timing_every) == 0),
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
void collect_spans(cmx_ctx *c) {  // call after the stream has been synchronised
  for (auto &s : c->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      c->t_ms[s.cls] += ms;
      c->t_n[s.cls] += 1;
    }
    c->event_pool.push_back(s.a);
    c->event_pool.push_back(s.b);
  }
  c->spans.clear();
}

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

int create_common(cmx_ctx **out, int kind, int device, int W, int H, const double *lut) {
  if (!out) return CMX_ERR_INVALID_ARG;
  *out = nullptr;
  if (W <= 0 || H <= 0 || W > 32767 || H > 32767 || !lut) return CMX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CMX_ERR_HIP;  // no CPU fallback
  if (device < 0 || device >= ndev) return CMX_ERR_INVALID_ARG;
  cmx_ctx *c = new cmx_ctx();
  c->kind = kind;
  c->device = device;
  c->W = W;
  c->H = H;
  *out = c;  // returned even on failure below so the caller can read cmx_last_error and destroy it
  HIP_TRY(c, hipSetDevice(device));
  HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  const size_t nl = (size_t)W * H * 3;
  HIP_TRY(c, hipMalloc((void **)&c->d_lut, nl * sizeof(double)));
  HIP_TRY(c, hipMemcpy(c->d_lut, lut, nl * sizeof(double), hipMemcpyHostToDevice));
  {  // image_geometry's rays are (x, y, 1): then the hot kernels read 16-byte (x, y) entries with one load
    const size_t npx = nl / 3;
    bool unit_z = true;
    for (size_t i = 0; i < npx && unit_z; i++) unit_z = lut[3 * i + 2] == 1.0;
    if (unit_z && npx > 0) {
      std::vector<double> xy(2 * npx);
      for (size_t i = 0; i < npx; i++) { xy[2 * i] = lut[3 * i]; xy[2 * i + 1] = lut[3 * i + 1]; }
      HIP_TRY(c, hipMalloc((void **)&c->d_lut2, 2 * npx * sizeof(double)));
      HIP_TRY(c, hipMemcpy(c->d_lut2, xy.data(), 2 * npx * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  c->result_cap = 4096;
  HIP_TRY(c, hipHostMalloc((void **)&c->h_result, c->result_cap * sizeof(double), hipHostMallocMapped));
  HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_result, c->h_result, 0));
  memset(c->h_result, 0, c->result_cap * sizeof(double));
  return CMX_OK;
}

int ensure_accum(cmx_ctx *c, size_t need);

// Fast path: swap to the partner buffer if it is known clean, otherwise clear the current one.  After this call
// c->d_accum is all-zero over `nplanes` planes and c->pingpong_planes tells the image pass to clear the partner.
int begin_accum(cmx_ctx *c, int nplanes, size_t np, bool fast) {
  c->pingpong_planes = 0;
  const size_t need = (size_t)nplanes * np;
  if (fast && !c->accum_external) {
    float *before = c->d_accum;
    int rc = ensure(c, c->d_accum, c->accum_cap, need);
    if (rc) return rc;
    if (c->d_accum != before) c->accum_clean = false;  // fresh allocation: contents undefined
    if (c->accum_alt_cap < need || !c->d_accum_alt) {
      rc = ensure(c, c->d_accum_alt, c->accum_alt_cap, c->accum_cap > need ? c->accum_cap : need);
      if (rc) return rc;
      c->alt_clean = false;
      c->alt_flagged = false;  // contents unknown: the next image pass clears every tile
    }
    if (c->alt_clean) {
      std::swap(c->d_accum, c->d_accum_alt);
      std::swap(c->accum_cap, c->accum_alt_cap);
      std::swap(c->accum_clean, c->alt_clean);
      std::swap(c->d_tflags, c->d_tflags_alt);
      std::swap(c->accum_flagged, c->alt_flagged);
    }
    if (!c->accum_clean) {
      Span sp(c, CMX_T_ZERO);
      HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, need * sizeof(float), c->stream));
      if (c->d_tflags) HIP_TRY(c, hipMemsetAsync(c->d_tflags, 0, c->tflags_cap, c->stream));
    }
    c->accum_flagged = false;  // set by the caller once a flag-marking splat has been launched into the clean buffer
    c->accum_clean = false;  // about to be written
    c->alt_clean = false;    // holds the previous evaluation's planes until this evaluation's image pass clears it
    c->pingpong_planes = nplanes;
    return CMX_OK;
  }
  int rc = ensure_accum(c, need);
  if (rc) return rc;
  {
    Span sp(c, CMX_T_ZERO);
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, need * sizeof(float), c->stream));
  }
  c->accum_clean = false;
  c->accum_flagged = false;
  return CMX_OK;
}

int ensure_accum(cmx_ctx *c, size_t need) {
  if (c->accum_external) {
    if (need > c->accum_cap)
      return fail(c, CMX_ERR_INVALID_ARG, "external accumulation buffer too small: %zu < %zu floats", c->accum_cap, need);
    return CMX_OK;
  }
  return ensure(c, c->d_accum, c->accum_cap, need);
}


// ---- sort the events by destination tile under the CURRENT parameters and build the chunk table
int do_binning(cmx_ctx *c, const FeSplatArgs *fe, const BeSplatArgs *be) {
  const int n = c->n_packed;
  const int W = c->imgW, H = c->imgH;
  const int tiles_x = (W + kBinTile - 1) / kBinTile, tiles_y = (H + kBinTile - 1) / kBinTile;
  const int planes_per_tile = fe ? 1 : 2;  // back end: key = 2*tile + (IL_new ? 1 : 0)
  const int ntiles = tiles_x * tiles_y * planes_per_tile;
  int rc;
  if ((size_t)n > c->bin_cap || !c->d_keys) {
    uint32_t **ptrs[6] = {&c->d_keys, &c->d_keys_s, &c->d_idx, &c->d_idx_s, &c->d_sxy, &c->d_sbatch};
    for (auto p : ptrs) {
      if (*p) HIP_TRY(c, hipFree(*p));
      *p = nullptr;
      HIP_TRY(c, hipMalloc((void **)p, (size_t)(n > 0 ? n : 1) * sizeof(uint32_t)));
    }
    c->bin_cap = (size_t)(n > 0 ? n : 1);
  }
  if (!c->d_fallback) {
    HIP_TRY(c, hipMalloc((void **)&c->d_fallback, sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(c->d_fallback, 0, sizeof(unsigned), c->stream));
  }
  rc = ensure(c, c->d_tile_start, c->tile_start_cap, (size_t)ntiles + 2);
  if (rc) return rc;
  // chunk size: hot tiles are split so that ~3 workgroups per CU exist; big chunks amortise the window flush
  int M = n / 768;
  M = M < 1536 ? 1536 : (M > 32768 ? 32768 : M);  // floor swept on MI355X (1M events: 1536 -> 11.8 us, 2048 -> 12.9, 1024 -> 15.3)
  M = (M + 255) / 256 * 256;
  // every tile contributes floor(len/M) full chunks and at most one remainder: an upper bound known on the host
  const int max_chunks = (n / M) + ntiles + 2;
  rc = ensure(c, c->d_chunks, c->chunks_cap, (size_t)max_chunks);
  if (rc) return rc;
  if (!c->d_nchunks) HIP_TRY(c, hipMalloc((void **)&c->d_nchunks, sizeof(int)));
  if (n > 0) {
    if (fe) launch_fe_bin_keys(*fe, tiles_x, ntiles, c->d_keys, c->d_idx, c->stream);
    else launch_be_bin_keys(*be, tiles_x, ntiles / 2, c->d_keys, c->d_idx, c->stream);
    int end_bit = 1;
    while ((1 << end_bit) <= ntiles) end_bit++;
    size_t tb = 0;
    if (sort_pairs_u32(nullptr, &tb, c->d_keys, c->d_keys_s, c->d_idx, c->d_idx_s, (unsigned)n, end_bit, c->stream) != 0)
      return fail(c, CMX_ERR_HIP, "rocprim radix sort (size query) failed");
    if (tb > c->sort_temp_cap) {
      if (c->d_sort_temp) HIP_TRY(c, hipFree(c->d_sort_temp));
      c->d_sort_temp = nullptr;
      HIP_TRY(c, hipMalloc(&c->d_sort_temp, tb));
      c->sort_temp_cap = tb;
    }
    if (sort_pairs_u32(c->d_sort_temp, &tb, c->d_keys, c->d_keys_s, c->d_idx, c->d_idx_s, (unsigned)n, end_bit, c->stream) != 0)
      return fail(c, CMX_ERR_HIP, "rocprim radix sort failed");
    launch_apply_perm(c->d_xy, c->d_idx_s, c->per_batch, n, c->d_sxy, c->d_sbatch, c->stream);
    launch_tile_lower_bound(c->d_keys_s, n, ntiles + 2, c->d_tile_start, c->stream);
    // the chunk table is built where the offsets are: no read-back, no host loop, no synchronisation
    launch_build_chunks(c->d_tile_start, ntiles, planes_per_tile, tiles_x, kBinMargin, M, c->d_chunks, c->d_nchunks, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->nchunks = max_chunks;
    c->nchunks_exact = false;
  } else {
    HIP_TRY(c, hipMemsetAsync(c->d_nchunks, 0, sizeof(int), c->stream));
    c->nchunks = 0;
    c->nchunks_exact = true;
  }
  c->bin_valid = true;
  c->rebin_count++;
  c->last_fallback_frac = 0;
  return CMX_OK;
}

BinnedEvents binned(const cmx_ctx *c) {
  BinnedEvents b{};
  b.sxy = c->d_sxy;
  b.sbatch = c->d_sbatch;
  b.chunks = c->d_chunks;
  b.nchunks = c->nchunks;
  b.nchunks_dev = c->d_nchunks;
  b.fallback = c->d_fallback;
  return b;
}

FeSplatArgs fe_args(const cmx_ctx *c, const double omega[3]) {
  FeSplatArgs a{};
  a.fx = c->fx; a.fy = c->fy; a.cx = c->cx; a.cy = c->cy;
  a.wx = omega[0]; a.wy = omega[1]; a.wz = omega[2];
  a.W = c->W; a.H = c->H;
  a.per_batch = c->per_batch;
  a.n = c->n_packed;
  a.xy = c->d_xy;
  a.batch_dt = c->d_batch_dt;
  a.lut = c->d_lut;
  a.lut2 = c->d_lut2;
  a.planes = c->d_accum;
  return a;
}

BeSplatArgs be_args(const cmx_ctx *c) {
  BeSplatArgs a{};
  a.W = c->W;
  a.Wp = c->Wp; a.Hp = c->Hp;
  a.fx = (double)((c->Wp / 360.0) * 180.0 / 3.1415926535897932384626433832795);  // focalFromFOV(.., 360, 180)
  a.fy = (double)((c->Hp / 180.0) * 180.0 / 3.1415926535897932384626433832795);
  a.cxp = (double)c->Wp / 2.0;
  a.cyp = (double)c->Hp / 2.0;
  a.per_batch = c->per_batch;
  a.n = c->n_packed;
  a.order = c->order;
  a.num_fixed = c->num_fixed;
  a.xy = c->d_xy;
  a.poseR = c->d_poseR;
  a.poses = c->d_poses;
  a.lut = c->d_lut;
  a.lut2 = c->d_lut2;
  a.planes = c->d_accum;
  return a;
}

bool adjoint_ok(const cmx_ctx *c) {  // G^T folding assumes single reflections: image larger than the kernel
  if (c->measure == CMX_GRADIENT_MAGNITUDE) return false;  // Sobel contrast: derivative-plane form only
  return c->grad_mode == CMX_GRAD_ADJOINT && c->imgW > 2 * c->radius + 1 && c->imgH > 2 * c->radius + 1;
}

// every evaluation ends in exactly one finalize launch; it carries the ticket sync_and_collect() waits for
void issue_finalize(cmx_ctx *c, FinalizeArgs &f, bool with_reduce) {
  f.ticket = ++c->ticket_issued;
  c->ticket_nout = 2 + (f.P > f.gP ? f.P : f.gP);
  if (with_reduce) launch_finalize(f, c->stream);
  else launch_finalize_only(f, c->stream);
}

// ping-pong partner clearing + tile-occupancy flags of an image pass (see ImgArgs)
int attach_tiles(cmx_ctx *c, ImgArgs &a, bool may_skip) {
  a.tiles_y = (a.H + kTileY - 1) / kTileY;
  if (a.zero_ptr) {
    if (c->alt_flagged && c->d_tflags_alt) {
      a.flags_other = c->d_tflags_alt;  // clear the dirty tiles only
    } else if (c->d_tflags_alt) {
      HIP_TRY(c, hipMemsetAsync(c->d_tflags_alt, 0, c->tflags_cap, c->stream));  // everything is cleared: no flag survives
    }
    c->alt_flagged = true;  // clean buffer, no flags: trivially consistent
  }
  if (may_skip && c->accum_flagged && c->d_tflags && (!a.igp || c->igp_flags_valid)) {
    a.flags_cur = c->d_tflags;
    a.flags_igp = a.igp ? c->d_igp_flags : nullptr;
  }
  return CMX_OK;
}

// large panoramas: compact the tiles that need work (call once a.partials / flags / zero_ptr are final)
int maybe_tile_list(cmx_ctx *c, ImgArgs &a, int reach) {
  if (!a.flags_cur || a.nblk <= kTileListMin) return CMX_OK;
  if ((size_t)a.nblk > c->tile_list_cap || !c->d_tile_list) {
    if (c->d_tile_list) HIP_TRY(c, hipFree(c->d_tile_list));
    if (c->d_tile_count) HIP_TRY(c, hipFree(c->d_tile_count));
    c->d_tile_list = c->d_tile_count = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_tile_list, (size_t)a.nblk * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc((void **)&c->d_tile_count, 2 * sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(c->d_tile_count, 0, 2 * sizeof(unsigned), c->stream));
    c->tile_list_cap = (size_t)a.nblk;
    c->tile_count_sel = 0;
  }
  unsigned *cur = c->d_tile_count + c->tile_count_sel, *next = c->d_tile_count + (c->tile_count_sel ^ 1);
  c->tile_count_sel ^= 1;  // this pass counts in `cur` and zeroes `next` for the pass after it
  launch_tile_list(a, reach, c->d_tile_list, cur, next, c->stream);
  a.tile_list = c->d_tile_list;
  a.tile_count = cur;
  return CMX_OK;
}

// image pass on the accumulated planes -> partial moments -> contrast/gradient in h_result
int run_image_and_finalize(cmx_ctx *c, int P, float *out_blur0, float *out_blurd) {
  const int W = c->imgW, H = c->imgH;
  const size_t np = (size_t)W * H;
  ImgArgs a{};
  a.W = W;
  a.H = H;
  a.r = c->radius;
  memcpy(a.taps, c->taps, sizeof(a.taps));
  if (c->kind == KIND_FE) {
    a.src_a = c->d_accum;
    a.src_b = nullptr;
    a.igp = nullptr;
    a.alpha = nullptr;
    a.dplanes = c->d_accum + np;
  } else {
    a.src_a = c->d_accum;
    a.src_b = c->d_accum + np;
    a.igp = c->ig_nonzero ? c->d_IGp : nullptr;
    a.alpha = c->d_alpha;
    a.dplanes = c->d_accum + 2 * np;
  }
  a.P = P;
  a.out_blur0 = out_blur0;
  a.out_blurd = out_blurd;
  if (c->pingpong_planes > 0 && c->d_accum_alt && !c->alt_clean) {
    a.zero_ptr = c->d_accum_alt;
    a.zero_planes = c->pingpong_planes;
    c->alt_clean = true;  // stream-ordered: clean by the time the next accumulate's splat runs
  }
  a.tiles_x = (W + kTileX - 1) / kTileX;
  a.nblk = a.tiles_x * ((H + kTileY - 1) / kTileY);
  const size_t nq = 2 + 2 * (size_t)P;
  int rc = attach_tiles(c, a, /*may_skip=*/P == 0 && !out_blur0 && !out_blurd);
  if (rc) return rc;
  rc = ensure(c, c->d_partials, c->partials_cap, nq * a.nblk);
  if (rc) return rc;
  rc = ensure(c, c->d_sums, c->sums_cap, nq);
  if (rc) return rc;
  if (2 + (size_t)P > c->result_cap - 1) return fail(c, CMX_ERR_INVALID_ARG, "too many derivative planes (%d)", P);
  a.partials = c->d_partials;
  if (c->measure == CMX_GRADIENT_MAGNITUDE && c->kind == KIND_FE) {
    // blurred planes -> scratch, then Sobel moments (reference local_focus_funcs.cpp:47-73), then finalize
    const size_t np = (size_t)W * H;
    float *blur = out_blur0;
    if (!blur) {
      rc = ensure(c, c->d_scratch, c->scratch_cap, 7 * np);
      if (rc) return rc;
      blur = c->d_scratch;
      a.out_blur0 = blur;
      a.out_blurd = blur + np;
    }
    SobelArgs sa{};
    sa.W = W; sa.H = H; sa.P = P;
    sa.planes = blur;
    sa.nblk = sobel_blocks(W, H);
    rc = ensure(c, c->d_gpartials, c->gpartials_cap, (size_t)(1 + P) * sa.nblk);
    if (rc) return rc;
    sa.partials = c->d_gpartials;
    Span sp(c, CMX_T_IMAGE);
    launch_image_moments(a, c->stream);
    launch_sobel_moments(sa, c->stream);
    FinalizeArgs f{};
    f.P = 0; f.nblk = a.nblk; f.measure = 2; f.npix = (double)np;
    f.partials = c->d_partials; f.sums = c->d_sums; f.result = c->d_result;
    f.direct = 1;
    f.gpartials = c->d_gpartials; f.gblocks = sa.nblk; f.gP = P;
    f.fallback = c->d_fallback;
    issue_finalize(c, f, false);
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  {
    Span sp(c, CMX_T_IMAGE);
    rc = maybe_tile_list(c, a, c->radius);
    if (rc) return rc;
    launch_image_moments(a, c->stream);
    FinalizeArgs f{};
    f.P = P;
    f.nblk = a.nblk;
    f.measure = c->measure;
    f.npix = (double)np;
    f.partials = c->d_partials;
    f.sums = c->d_sums;
    f.result = c->d_result;
    f.fallback = c->d_fallback;
    if (P == 0 && (a.nblk <= 2048 || a.tile_list)) {
      f.direct = 1;
      f.nvalid = a.tile_count;  // list path: partial rows are compact, one entry per listed tile
      issue_finalize(c, f, false);
    } else {
      issue_finalize(c, f, true);
    }
  }
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}


// adjoint gradient: fused image pass (B = G*A with its moments, Jt = G^T B^) -> gather over the events (S1, and S2 for
// the votes next to the border) -> finalize: contrast from the moments, grad = (2/N)(S1 - mu*S2).
// phase 0 = everything; 1 = up to the per-rank partial sums (d_gsum, 2P doubles); 2 = finalize from d_gsum.
int run_adjoint(cmx_ctx *c, int P, int phase = 0) {
  const int W = c->imgW, H = c->imgH;
  const size_t np = (size_t)W * H;
  float *jt_before = c->d_itilde;
  int rc = ensure(c, c->d_itilde, c->itilde_cap, np);
  if (rc) return rc;
  if (c->d_itilde != jt_before)  // tiles the image pass skips keep whatever they held: make that finite from the start
    HIP_TRY(c, hipMemsetAsync(c->d_itilde, 0, c->itilde_cap * sizeof(float), c->stream));
  ImgAdjArgs ia{};
  ImgArgs &a = ia.img;
  a.W = W; a.H = H; a.r = c->radius;
  memcpy(a.taps, c->taps, sizeof(a.taps));
  a.src_a = c->d_accum;
  if (c->kind == KIND_BE) {
    a.src_b = c->d_accum + np;
    a.igp = c->ig_nonzero ? c->d_IGp : nullptr;
    a.alpha = c->d_alpha;
  }
  a.P = 0;
  a.tiles_x = image_adjoint_tiles_x(W);
  a.nblk = image_adjoint_tiles(W, H);
  if (phase != 2 && c->pingpong_planes > 0 && c->d_accum_alt && !c->alt_clean) {
    a.zero_ptr = c->d_accum_alt;
    a.zero_planes = c->pingpong_planes;
    c->alt_clean = true;
  }
  rc = attach_tiles(c, a, /*may_skip=*/true);
  if (rc) return rc;
  ia.jt = c->d_itilde;
  rc = ensure(c, c->d_partials, c->partials_cap, (size_t)2 * a.nblk);
  if (rc) return rc;
  rc = ensure(c, c->d_sums, c->sums_cap, 2);
  if (rc) return rc;
  // rows of the gather partial table: front end = gather workgroups; back end = workgroups of the per-batch pass
  const int gb = (c->kind == KIND_FE) ? fe_gather_blocks(c->n_packed) : be_batch_blocks(c->nb);
  const int P2 = 2 * (P > 0 ? P : 1);
  rc = ensure(c, c->d_gpartials, c->gpartials_cap, (size_t)gb * P2);
  if (rc) return rc;
  // four events per lane in the back-end gather when a lane's four events cannot straddle a batch boundary
  const int slice_shift = (c->kind == KIND_BE && c->per_batch % 4 == 0) ? 8 : 6;
  const int parts_per_batch = ((c->per_batch + (1 << slice_shift) - 2) >> slice_shift) + 1;
  if (c->kind == KIND_BE) {
    rc = ensure(c, c->d_vparts, c->vparts_cap, (size_t)(c->nb > 0 ? c->nb : 1) * parts_per_batch * 6);
    if (rc) return rc;
  }
  if (2 + (size_t)P > c->result_cap - 2) return fail(c, CMX_ERR_INVALID_ARG, "too many parameters (%d)", P);
  if (!c->gsum_external) {
    rc = ensure(c, c->d_gsum, c->gsum_cap, (size_t)P2);
    if (rc) return rc;
  } else if ((size_t)P2 > c->gsum_cap) {
    return fail(c, CMX_ERR_INVALID_ARG, "external gradient buffer too small: %zu < %d doubles", c->gsum_cap, P2);
  }
  a.partials = c->d_partials;
  FinalizeArgs f{};
  f.P = 0;
  f.nblk = a.nblk;
  f.measure = c->measure;
  f.npix = (double)np;
  f.partials = c->d_partials;
  f.sums = c->d_sums;
  f.result = c->d_result;
  if (phase != 2) {  // large panoramas: compact work list (a pre-pass kernel; partial rows become compact too)
    rc = maybe_tile_list(c, a, 2 * c->radius);
    if (rc) return rc;
  }
  const bool direct = a.nblk <= 2048 || a.tile_list;  // few entries: finalize sums the per-tile moments itself
  f.direct = direct ? 1 : 0;
  f.nvalid = a.tile_count;
  f.mu_free = 1;
  if (phase == 0) {  // single call: finalize sums the gather kernel's block partials itself
    f.gpartials = c->d_gpartials;
    f.gblocks = gb;
  } else {           // split call: finalize reads the (all-reduced) per-parameter sums
    f.gpartials = c->d_gsum;
    f.gblocks = 1;
  }
  f.gP = P;
  f.fallback = c->d_fallback;
  if (phase == 2) {
    issue_finalize(c, f, false);
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  {
    Span sp(c, CMX_T_IMAGE);
    launch_image_adjoint(ia, c->stream);
    if (!direct) launch_reduce_partials(f, c->stream);
  }
  {
    Span sp(c, CMX_T_GATHER, /*exact=*/true);
    if (c->kind == KIND_FE) {
      FeGatherArgs g{};
      g.ev = fe_args(c, c->last_x);
      g.itilde = c->d_itilde;
      g.gpartials = c->d_gpartials;
      g.cx = c->d_cx; g.cy = c->d_cy; g.r = c->radius;
      if (c->splat_mode == 1 && c->bin_valid) {  // tile order: the same sorted arrays the LDS splat consumes
        g.sxy = c->d_sxy;
        g.sbatch = c->d_sbatch;
      }
      if (c->n_packed > 0) launch_fe_gather(g, c->stream, sp.t0(), sp.t1());
      else HIP_TRY(c, hipMemsetAsync(c->d_gpartials, 0, (size_t)gb * P2 * sizeof(double), c->stream));
    } else {
      BeGatherArgs g{};
      g.ev = be_args(c);
      g.itilde = c->d_itilde;
      g.P = P;
      g.gpartials = c->d_gpartials;
      g.cx = c->d_cx; g.cy = c->d_cy; g.r = c->radius;
      g.vparts = c->d_vparts;
      g.parts_per_batch = parts_per_batch;
      g.slice_shift = slice_shift;
      if (c->n_packed > 0 && P > 0) launch_be_gather(g, c->nb, c->stream, sp.t0(), sp.t1());
      else HIP_TRY(c, hipMemsetAsync(c->d_gpartials, 0, (size_t)gb * P2 * sizeof(double), c->stream));
    }
    if (phase == 1) launch_reduce_gpartials(c->d_gpartials, gb, 2 * P, c->d_gsum, c->stream);
  }
  if (phase == 1) {
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  issue_finalize(c, f, false);
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}

int sync_and_collect(cmx_ctx *c, bool ends_in_finalize = false) {
  // ends_in_finalize: the last thing queued on the stream is an evaluation's finalize kernel.  Wait for it through
  // its completion ticket in mapped host memory (a few microseconds earlier than the runtime reports the stream idle);
  // anything slower than the spin budget, and every caller that queued copies or other kernels after the finalize,
  // takes the ordinary stream synchronisation.
  bool done = false;
  if (ends_in_finalize && c->ticket_wait && c->ticket_issued) {
    const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(c->h_result);
    const unsigned long long want = c->ticket_issued;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
      if (w[kTicketSlot] == want) {  // ticket seen: accept only a consistent snapshot of the results
        unsigned long long x = w[kFallbackSlot];
        for (int k = 0; k < c->ticket_nout; k++) x ^= w[k];
        if ((x ^ (want * kTicketMix)) == w[kChecksumSlot]) { done = true; break; }
      }
      __builtin_ia32_pause();
      if ((spins & 1023u) == 1023u &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0)
        break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done) HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!c->nchunks_exact && c->bin_valid && c->d_nchunks) {  // once per binning: launch exactly the chunks that exist
    int nch = 0;
    if (hipMemcpy(&nch, c->d_nchunks, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && nch >= 0 && nch <= c->nchunks) c->nchunks = nch;
    c->nchunks_exact = true;
  }
  if (c->fallback_pending && c->n_packed > 0) c->last_fallback_frac = c->h_result[kFallbackSlot] / (double)c->n_packed;
  c->fallback_pending = false;
  // timing spans are resolved lazily (cmx_timing_get) so that timed evaluations wait exactly like untimed ones
  if (c->spans.size() > 4096) {
    if (done) HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_spans(c);
  }
  return CMX_OK;
}

// split [0, n) over a few host threads (the AoS->SoA packing of millions of events is memory-bound on one core).
// The workers are created once per process and parked on a condition variable: spawning eight threads per call costs
// more than the packing they do (0.25 ms per call against ~0.1 ms of work for a 1M-event packet).
class HostPool {
 public:
  static HostPool &get() {
    static HostPool p;
    return p;
  }
  int workers() const { return (int)th_.size(); }
  // run job(k) for k in [0, parts) on the workers and the caller; returns when all are done.  One caller at a time
  // per process is enough here (packing is a fraction of a millisecond), so concurrent callers serialise.
  void run(int parts, const std::function<void(int)> &job) {
    std::lock_guard<std::mutex> serial(run_mutex_);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      next_ = 0;
      parts_ = parts;
      pending_ = parts;
      generation_++;
    }
    cv_.notify_all();
    work();  // the caller takes parts too
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  HostPool() {
    unsigned hw = std::thread::hardware_concurrency();
    const int T = (int)(hw ? (hw > 8 ? 8 : hw) : 1);
    for (int t = 1; t < T; t++) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  void work() {
    for (;;) {
      int k;
      const std::function<void(int)> *job;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!job_ || next_ >= parts_) return;
        k = next_++;
        job = job_;
      }
      (*job)(k);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, run_mutex_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  int next_ = 0, parts_ = 0, pending_ = 0;
  unsigned long long generation_ = 0;
  bool stop_ = false;
};

template <typename F>
void parallel_ranges(int64_t n, F fn, int64_t serial_below = 262144) {
  int T = HostPool::get().workers() + 1;
  if (n < serial_below) T = 1;
  if (T <= 1) { fn((int64_t)0, n); return; }
  const int64_t per = (n + T - 1) / T;
  const int parts = (int)((n + per - 1) / per);
  HostPool::get().run(parts, [&](int k) {
    const int64_t a = (int64_t)k * per, b = (a + per < n) ? a + per : n;
    if (a < b) fn(a, b);
  });
}

// argument checks only; the coordinate range is validated inside the packing pass (one sweep over the events instead
// of two) and, if that pass saw an offender, located by check_events below
int check_event_args(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t) {
  if (n < 0 || n > kMaxEvents) return fail(c, CMX_ERR_INVALID_ARG, "bad event count %lld (limit %lld)", (long long)n, (long long)kMaxEvents);
  if (n > 0 && (!x || !y || !t)) return fail(c, CMX_ERR_INVALID_ARG, "null event arrays");
  return CMX_OK;
}
int check_events(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t) {
  int rc0 = check_event_args(c, n, x, y, t);
  if (rc0) return rc0;
  const int W = c->W, H = c->H;
  std::atomic<int64_t> bad(-1);
  parallel_ranges(n, [&](int64_t a, int64_t b) {
    unsigned acc = 0;
    for (int64_t i = a; i < b; i++) acc |= (unsigned)(x[i] >= W) | (unsigned)(y[i] >= H);
    if (acc)
      for (int64_t i = a; i < b; i++)
        if (x[i] >= W || y[i] >= H) {
          int64_t cur = bad.load();
          while ((cur < 0 || i < cur) && !bad.compare_exchange_weak(cur, i)) {}
          break;
        }
  });
  const int64_t i = bad.load();
  if (i >= 0)
    return fail(c, CMX_ERR_EVENT_RANGE, "event %lld at (%u,%u) outside the %dx%d sensor", (long long)i, x[i], y[i], W, H);
  return CMX_OK;
}

int ensure_pinned_xy(cmx_ctx *c, size_t n) {
  if (n <= c->h_xy_cap && c->h_xy) return CMX_OK;
  if (c->h_xy) HIP_TRY(c, hipHostFree(c->h_xy));
  c->h_xy = nullptr;
  c->h_xy_cap = 0;
  const size_t cap = n + n / 4 + 1024;
  HIP_TRY(c, hipHostMalloc((void **)&c->h_xy, cap * sizeof(uint32_t), hipHostMallocDefault));
  c->h_xy_cap = cap;
  return CMX_OK;
}

}  // namespace

// =============================================================================================== generic
extern "C" {

const char *cmx_version(void) { return "cmax-hip 0.1 (gfx950)"; }

int cmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *cmx_last_error(const cmx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char *cmx_status_string(int s) {
  switch (s) {
    case CMX_OK: return "ok";
    case CMX_ERR_INVALID_ARG: return "invalid argument";
    case CMX_ERR_EVENT_RANGE: return "event coordinates outside the sensor";
    case CMX_ERR_HIP: return "HIP runtime error (no GPU / launch failure)";
    case CMX_ERR_SPLINE_RANGE: return "batch time outside the spline's knot support";
    case CMX_ERR_STATE: return "call sequence error";
    case CMX_ERR_TIME_ORDER: return "event batch spans a negative time interval";
    default: return "unknown status";
  }
}

void cmx_destroy(cmx_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto &s : c->spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
  for (auto e : c->event_pool) hipEventDestroy(e);
  hipFree(c->d_lut);
  hipFree(c->d_lut2);
  hipFree(c->d_nchunks);
  hipFree(c->d_batch_err);
  hipFree(c->d_xy);
  if (c->h_xy) hipHostFree(c->h_xy);
  hipFree(c->d_batch_dt);
  hipFree(c->d_batch_t);
  hipFree(c->d_poses);
  hipFree(c->d_poseR);
  if (c->h_spline) hipHostFree(c->h_spline);
  hipFree(c->d_IG);
  hipFree(c->d_visits);
  hipFree(c->d_mask);
  hipFree(c->d_IGp);
  hipFree(c->d_alpha);
  if (!c->accum_external) hipFree(c->d_accum);
  hipFree(c->d_accum_alt);
  hipFree(c->d_scratch);
  hipFree(c->d_partials);
  hipFree(c->d_sums);
  hipFree(c->d_keys); hipFree(c->d_keys_s); hipFree(c->d_idx); hipFree(c->d_idx_s); hipFree(c->d_sxy); hipFree(c->d_sbatch);
  hipFree(c->d_sort_temp);
  hipFree(c->d_tile_start);
  hipFree(c->d_chunks);
  hipFree(c->d_fallback);
  hipFree(c->d_itilde);
  hipFree(c->d_cx);
  hipFree(c->d_cy);
  hipFree(c->d_gpartials);
  hipFree(c->d_tflags); hipFree(c->d_tflags_alt); hipFree(c->d_igp_flags);
  hipFree(c->d_tile_list); hipFree(c->d_tile_count);
  hipFree(c->d_vparts);
  if (!c->gsum_external) hipFree(c->d_gsum);
  if (c->h_result) hipHostFree(c->h_result);
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int cmx_set_option(cmx_ctx *c, int key, int value) {
  if (!c) return CMX_ERR_INVALID_ARG;
  switch (key) {
    case CMX_OPT_GRAD_MODE:
      if (value != CMX_GRAD_PLANES && value != CMX_GRAD_ADJOINT) return fail(c, CMX_ERR_INVALID_ARG, "bad grad mode %d", value);
      c->grad_mode = value;
      return CMX_OK;
    case CMX_OPT_SPLAT_MODE:
      if (value != 0 && value != 1) return fail(c, CMX_ERR_INVALID_ARG, "bad splat mode %d", value);
      c->splat_mode = value;
      c->bin_valid = false;
      return CMX_OK;
    case CMX_OPT_REUSE_IMAGE:
      c->reuse_image = value != 0;
      return CMX_OK;
    case CMX_OPT_SPIN_WAIT:
      c->ticket_wait = value != 0;
      return CMX_OK;
    default: return fail(c, CMX_ERR_INVALID_ARG, "unknown option %d", key);
  }
}

int cmx_set_stream(cmx_ctx *c, void *hip_stream) {
  if (!c) return CMX_ERR_INVALID_ARG;
  int rc = bind(c);
  if (rc) return rc;
  if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) { HIP_TRY(c, hipStreamDestroy(c->stream)); c->stream = nullptr; c->own_stream = false; }
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
  } else {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return CMX_OK;
}

int cmx_get_stats(cmx_ctx *c, double stats[8]) {
  if (!c || !stats) return CMX_ERR_INVALID_ARG;
  stats[0] = (double)c->rebin_count;
  stats[1] = c->last_fallback_frac;
  {  // true length of the chunk table (device-resident until the first evaluation after a binning has been collected)
    int nch = c->nchunks;
    if (!c->nchunks_exact && c->d_nchunks && c->bin_valid && bind(c) == CMX_OK && hipStreamSynchronize(c->stream) == hipSuccess)
      (void)hipMemcpy(&nch, c->d_nchunks, sizeof(int), hipMemcpyDeviceToHost);
    stats[2] = (double)nch;
  }
  stats[3] = (double)c->n_packed;
  stats[4] = (double)c->reuse_hits;
  return CMX_OK;
}

int cmx_timing_enable(cmx_ctx *c, int on) {
  if (!c) return CMX_ERR_INVALID_ARG;
  c->timing = (on & 0xff) != 0;
  c->timing_mask = on & 0xff;
  c->timing_every = (on >> 8) > 0 ? (on >> 8) : 1;  // bits 8..: sample every n-th evaluation only
  c->timing_tick = 0;
  return CMX_OK;
}
int cmx_timing_get(cmx_ctx *c, double ms[CMX_T_COUNT], int64_t launches[CMX_T_COUNT]) {
  if (!c) return CMX_ERR_INVALID_ARG;
  if (!c->spans.empty()) {
    int rc = bind(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_spans(c);
  }
  for (int i = 0; i < CMX_T_COUNT; i++) {
    if (ms) ms[i] = c->t_ms[i];
    if (launches) launches[i] = c->t_n[i];
    c->t_ms[i] = 0;
    c->t_n[i] = 0;
  }
  return CMX_OK;
}

size_t cmx_accum_capacity(const cmx_ctx *c) {
  if (!c) return 0;
  if (c->kind == KIND_FE) return (size_t)4 * c->W * c->H;
  const int P = 3 * (c->K - c->num_fixed);
  return (size_t)(2 + (P > 0 ? P : 0)) * c->Wp * c->Hp;
}
int cmx_set_accum_buffer(cmx_ctx *c, void *device_ptr, size_t n_floats) {
  if (!c) return CMX_ERR_INVALID_ARG;
  int rc = bind(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!c->accum_external && c->d_accum) HIP_TRY(c, hipFree(c->d_accum));
  c->d_accum = (float *)device_ptr;
  c->accum_cap = device_ptr ? n_floats : 0;
  c->accum_external = device_ptr != nullptr;
  c->accumulated = false;
  return CMX_OK;
}
void *cmx_accum_ptr(const cmx_ctx *c) { return c ? c->d_accum : nullptr; }
size_t cmx_accum_count(const cmx_ctx *c) { return c ? c->accum_count : 0; }

int64_t cmx_traj_temp_start_ns(double t_beg, int idx_traj_beg, double dt_knots) {
  const double t = t_beg + idx_traj_beg * dt_knots;
  return (int64_t)(1e9 * t);
}

// =============================================================================================== front end
int cmx_frontend_create(cmx_ctx **out, int device, int W, int H, const double *lut) {
  int rc = create_common(out, KIND_FE, device, W, H, lut);
  if (rc) return rc;
  (*out)->imgW = W;
  (*out)->imgH = H;
  return CMX_OK;
}

// d_raw != nullptr: the events are already on the device (event store), x / y are unused and t_ns is the store's
// host mirror of the timestamps
static int fe_set_packet_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                              const uint32_t *d_raw, int64_t t_ref_ns, double fx, double fy, double cx, double cy,
                              int event_batch_size, double blur_sigma, int contrast_measure) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  int rc = bind(c);
  if (rc) return rc;
  c->have_data = false;
  c->accumulated = false;
  c->x_valid = false;
  if (event_batch_size <= 0) return fail(c, CMX_ERR_INVALID_ARG, "event_batch_size must be > 0");
  // computeContrast's switch (local_focus_funcs.cpp:98-109): 1 = mean square, 2 = gradient magnitude, default = variance
  if (contrast_measure != CMX_MEAN_SQUARE && contrast_measure != CMX_GRADIENT_MAGNITUDE) contrast_measure = CMX_VARIANCE;
  if (!d_raw) {
    rc = check_event_args(c, n, x, y, t_ns);
    if (rc) return rc;
  } else if (n < 0 || n > kMaxEvents) {
    return fail(c, CMX_ERR_INVALID_ARG, "bad event count %lld", (long long)n);
  }
  rc = setup_blur(c, blur_sigma);
  if (rc) return rc;
  c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy;
  c->batch = event_batch_size;
  c->measure = contrast_measure;

  // SoA packing + per-batch dt = time_batch.toSec() - time_ref.toSec()  (local_image_warped_events.cpp:68-75)
  const int nb = (int)((n + event_batch_size - 1) / event_batch_size);
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the pinned staging buffer may still feed the previous upload
  uint32_t *xy = nullptr;
  if (!d_raw) {
    rc = ensure_pinned_xy(c, (size_t)n);
    if (rc) return rc;
    xy = c->h_xy;
    std::atomic<unsigned> out_of_range(0);
    const unsigned W = (unsigned)c->W, H = (unsigned)c->H;
    parallel_ranges(n, [&](int64_t a, int64_t b) {
      unsigned acc = 0;
      for (int64_t i = a; i < b; i++) {
        acc |= (unsigned)(x[i] >= W) | (unsigned)(y[i] >= H);
        xy[i] = (uint32_t)x[i] | ((uint32_t)y[i] << 16);
      }
      if (acc) out_of_range = 1;
    });
    if (out_of_range.load()) return check_events(c, n, x, y, t_ns);  // locate and report the offender
  }
  std::vector<double> dts((size_t)nb);
  const double tref = time_to_sec(t_ref_ns);
  std::atomic<int> bad_batch(-1);
  parallel_ranges(nb, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      const int64_t beg = b * event_batch_size;
      const int64_t end = (beg + event_batch_size < n) ? beg + event_batch_size : n;
      if (t_ns[end - 1] < t_ns[beg]) { bad_batch = (int)b; return; }
      dts[(size_t)b] = time_to_sec(time_batch_ns(t_ns[beg], t_ns[end - 1])) - tref;
    }
  }, /*serial_below=*/4096);
  if (bad_batch.load() >= 0) return fail(c, CMX_ERR_TIME_ORDER, "batch %d spans a negative time interval", bad_batch.load());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  rc = ensure(c, c->d_xy, c->xy_cap, (size_t)n);
  if (rc) return rc;
  rc = ensure(c, c->d_batch_dt, c->batch_cap, (size_t)nb);
  if (rc) return rc;
  if (n) {
    if (d_raw) HIP_TRY(c, hipMemcpyAsync(c->d_xy, d_raw, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(c, hipMemcpyAsync(c->d_xy, xy, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpy(c->d_batch_dt, dts.data(), (size_t)nb * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->n_packed = (int)n;
  c->per_batch = event_batch_size;
  c->nb = nb;
  c->have_data = true;
  c->bin_valid = false;
  return CMX_OK;
}

int cmx_frontend_set_packet(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                            int64_t t_ref_ns, double fx, double fy, double cx, double cy, int event_batch_size,
                            double blur_sigma, int contrast_measure) {
  return fe_set_packet_impl(c, n, x, y, t_ns, nullptr, t_ref_ns, fx, fy, cx, cy, event_batch_size, blur_sigma, contrast_measure);
}

static int fe_accumulate(cmx_ctx *c, const double omega[3], int nplanes) {
  c->timing_tick++;  // every span of this evaluation (accumulate and finish) samples, or none does
  const size_t np = (size_t)c->W * c->H;
  int rc = begin_accum(c, nplanes, np, nplanes == 1 && adjoint_ok(c) && c->splat_mode == 1);
  if (rc) return rc;
  FeSplatArgs a = fe_args(c, omega);
  for (int k = 0; k < 3; k++) c->last_x[k] = omega[k];
  const bool use_lds = c->splat_mode == 1 && nplanes == 1 && c->n_packed > 0;
  if (use_lds && (!c->bin_valid || c->last_fallback_frac > 0.15)) {
    rc = do_binning(c, &a, nullptr);
    if (rc) return rc;
  }
  {
    Span sp(c, CMX_T_SPLAT, /*exact=*/true);
    c->last_used_lds = use_lds;
    if (use_lds) c->fallback_pending = true;
    if (use_lds) launch_fe_splat_lds(a, binned(c), c->stream, sp.t0(), sp.t1());
    else launch_fe_splat(a, nplanes > 1, c->stream, sp.t0(), sp.t1());
  }
  HIP_TRY(c, hipGetLastError());
  c->accum_count = nplanes * np;
  c->last_P = nplanes - 1;
  c->accumulated = true;
  c->x_valid = true;
  return CMX_OK;
}

int cmx_frontend_accumulate(cmx_ctx *c, const double omega[3], int want_grad) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_frontend_set_packet has not succeeded");
  if (!omega) return fail(c, CMX_ERR_INVALID_ARG, "null omega");
  int rc = bind(c);
  if (rc) return rc;
  c->last_adjoint = want_grad && adjoint_ok(c);
  return fe_accumulate(c, omega, (want_grad && !c->last_adjoint) ? 4 : 1);
}

int cmx_frontend_finish(cmx_ctx *c, double *contrast, double *grad) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish without accumulate");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  if (grad && c->last_P != 3 && !c->last_adjoint)
    return fail(c, CMX_ERR_STATE, "gradient requested but accumulate ran without it");
  int rc = bind(c);
  if (rc) return rc;
  if (grad && c->last_adjoint) rc = run_adjoint(c, 3);
  else rc = run_image_and_finalize(c, grad ? 3 : 0, nullptr, nullptr);
  if (rc) return rc;
  rc = sync_and_collect(c, true);
  if (rc) return rc;
  *contrast = c->h_result[0];
  if (grad) for (int k = 0; k < 3; k++) grad[k] = c->h_result[2 + k];
  return CMX_OK;
}

static bool can_reuse(const cmx_ctx *c, const double *x, int n, bool want_grad) {
  if (!want_grad || !c->reuse_image || !c->have_data || !c->accumulated || !c->x_valid) return false;
  if (!adjoint_ok(c) || c->accum_external) return false;
  return memcmp(x, c->last_x, sizeof(double) * (size_t)n) == 0;
}

// evaluation with an attached communicator: the two exchange points of SURVEY.md section 8e, in place, on the stream
static int finish_begin(cmx_ctx *c, int kind, int want_grad);
static int finish_end(cmx_ctx *c, int kind, double *contrast, double *grad);
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

static int finish_sharded(cmx_ctx *c, int kind, bool exchange_planes, double *contrast, double *grad) {
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

int cmx_frontend_eval(cmx_ctx *c, const double omega[3], double *contrast, double *grad) {
  const bool sharded = c && c->comm;
  if (c && c->kind == KIND_FE && omega && can_reuse(c, omega, 3, grad != nullptr)) {
    c->last_adjoint = true;  // image of this very point is resident: adjoint blur + gather only
    c->reuse_hits++;
    if (sharded) return finish_sharded(c, KIND_FE, false, contrast, grad);
    return cmx_frontend_finish(c, contrast, grad);
  }
  int rc = cmx_frontend_accumulate(c, omega, grad != nullptr);
  if (rc) return rc;
  if (sharded) return finish_sharded(c, KIND_FE, true, contrast, grad);
  return cmx_frontend_finish(c, contrast, grad);
}

int cmx_frontend_get_iwe(cmx_ctx *c, const double omega[3], int blur, float *iwe, float *deriv) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_frontend_set_packet has not succeeded");
  if (!omega || !iwe) return fail(c, CMX_ERR_INVALID_ARG, "null argument");
  int rc = bind(c);
  if (rc) return rc;
  const size_t np = (size_t)c->W * c->H;
  const int nplanes = deriv ? 4 : 1;
  c->last_adjoint = false;
  rc = fe_accumulate(c, omega, nplanes);
  c->x_valid = false;
  if (rc) return rc;
  rc = ensure(c, c->d_scratch, c->scratch_cap, 7 * np);
  if (rc) return rc;
  const float *src = c->d_accum;
  if (blur && c->radius > 0) {
    rc = run_image_and_finalize(c, nplanes - 1, c->d_scratch, c->d_scratch + np);
    if (rc) return rc;
    src = c->d_scratch;
  }
  HIP_TRY(c, hipMemcpyAsync(iwe, src, np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (deriv) {
    float *inter = c->d_scratch + 4 * np;
    launch_interleave3(src + np, inter, (int)np, c->stream);
    HIP_TRY(c, hipMemcpyAsync(deriv, inter, 3 * np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  return sync_and_collect(c);
}

// =============================================================================================== back end
int cmx_backend_create(cmx_ctx **out, int device, int W, int H, const double *lut, int Wp, int Hp) {
  if (Wp < 4 || Hp < 4 || Wp > 65535 || Hp > 65535) { if (out) *out = nullptr; return CMX_ERR_INVALID_ARG; }
  int rc = create_common(out, KIND_BE, device, W, H, lut);
  if (rc) return rc;
  cmx_ctx *c = *out;
  c->Wp = Wp; c->Hp = Hp;
  c->imgW = Wp; c->imgH = Hp;
  const size_t np = (size_t)Wp * Hp;
  HIP_TRY(c, hipMalloc((void **)&c->d_IG, np * sizeof(float)));
  HIP_TRY(c, hipMalloc((void **)&c->d_IGp, np * sizeof(float)));
  HIP_TRY(c, hipMemset(c->d_IG, 0, np * sizeof(float)));
  HIP_TRY(c, hipMemset(c->d_IGp, 0, np * sizeof(float)));
  HIP_TRY(c, hipMalloc((void **)&c->d_visits, np));
  HIP_TRY(c, hipMalloc((void **)&c->d_mask, np));
  HIP_TRY(c, hipMemset(c->d_visits, 0, np));
  HIP_TRY(c, hipMemset(c->d_mask, 0, np));
  HIP_TRY(c, hipMalloc((void **)&c->d_alpha, sizeof(double)));
  HIP_TRY(c, hipMemset(c->d_alpha, 0, sizeof(double)));
  HIP_TRY(c, hipHostMalloc((void **)&c->h_spline, sizeof(SplineArgs), hipHostMallocDefault));
  return CMX_OK;
}

static int be_set_window_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                              const uint32_t *d_raw, const int64_t *d_t, int order, int K, const double *knots_xyzw,
                              int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                              int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure,
                              const float *IG) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind(c);
  if (rc) return rc;
  c->have_data = false;
  c->accumulated = false;
  c->x_valid = false;
  if (order != 2 && order != 4) return fail(c, CMX_ERR_INVALID_ARG, "spline order %d unsupported (2 = linear, 4 = cubic)", order);
  if (K < order || K > kMaxKnots) return fail(c, CMX_ERR_INVALID_ARG, "K=%d outside [%d, %d]", K, order, kMaxKnots);
  if (num_fixed < 0 || num_fixed > K) return fail(c, CMX_ERR_INVALID_ARG, "num_fixed=%d outside [0, K]", num_fixed);
  if (!knots_xyzw || dt_ns <= 0) return fail(c, CMX_ERR_INVALID_ARG, "bad spline description");
  if (event_batch_size <= 0 || event_sample_rate <= 0) return fail(c, CMX_ERR_INVALID_ARG, "batch size / sample rate must be > 0");
  // the back end's switch (global_focus_funcs.cpp:61-69) knows mean square only; everything else is variance
  if (contrast_measure != CMX_MEAN_SQUARE) contrast_measure = CMX_VARIANCE;
  if (!d_raw) {
    rc = check_event_args(c, n, x, y, t_ns);
    if (rc) return rc;
  } else if (n < 0 || n > kMaxEvents) {
    return fail(c, CMX_ERR_INVALID_ARG, "bad event count %lld", (long long)n);
  }
  rc = setup_blur(c, blur_sigma);
  if (rc) return rc;

  // Batches: for (beg = 0; beg < n-1; beg += B) { end = (n-beg > B) ? beg+B : n; }  -- a trailing batch holding
  // exactly the last single event is skipped (event_pano_warper.cpp:188-196); inside a batch events are taken
  // with stride event_sample_rate restarting at the batch start (:262).
  const int B = event_batch_size, rate = event_sample_rate;
  const int per_batch = (B + rate - 1) / rate;
  const int64_t nb64 = (n > 1) ? (n - 1 + B - 1) / B : 0;
  if (nb64 > 0x7fffffffLL) return fail(c, CMX_ERR_INVALID_ARG, "too many batches");
  const int nbatches = (int)nb64;
  int64_t n_packed_total = 0;
  if (nbatches > 0) {
    const int64_t last_beg = (int64_t)(nbatches - 1) * B;
    const int64_t last_len = (n - last_beg > B) ? B : (n - last_beg);  // a trailing single event is never in a batch
    n_packed_total = (int64_t)(nbatches - 1) * per_batch + (last_len + rate - 1) / rate;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the pinned staging buffer may still feed the previous upload
  uint32_t *xy = nullptr;
  if (!d_raw) {
    rc = ensure_pinned_xy(c, (size_t)n_packed_total);
    if (rc) return rc;
    xy = c->h_xy;
  }
  std::vector<long long> bt(d_raw ? 0 : (size_t)nbatches);
  std::atomic<unsigned> out_of_range(0);  // some event outside the sensor: seen by the packing pass, located afterwards
  const unsigned sensor_w = (unsigned)c->W, sensor_h = (unsigned)c->H;
  std::atomic<int> err_kind(0);
  std::atomic<long long> err_at(-1);
  if (!d_raw)  // (windows cut from the event store get their batch times from a kernel, below)
  parallel_ranges(nbatches, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      const int64_t beg = b * B;
      const int64_t end = (n - beg > B) ? beg + B : n;
      if (t_ns[end - 1] < t_ns[beg]) { err_kind = CMX_ERR_TIME_ORDER; err_at = beg; return; }
      const long long tb = time_batch_ns(t_ns[beg], t_ns[end - 1]);
      const long long st = tb - start_ns;
      if (st < 0 || st / dt_ns + order > K) { err_kind = CMX_ERR_SPLINE_RANGE; err_at = tb; return; }
      bt[(size_t)b] = tb;
      if (rate == 1) continue;  // packed below by a flat, vectorisable loop (packed index == event index)
      uint32_t *dst = xy + b * per_batch;
      unsigned acc = 0;
      for (int64_t e = beg; e < end; e += rate) {
        acc |= (unsigned)(x[e] >= sensor_w) | (unsigned)(y[e] >= sensor_h);
        *dst++ = (uint32_t)x[e] | ((uint32_t)y[e] << 16) | ((t_ns[e] < t_next_win_beg_ns) ? 0x80000000u : 0u);
      }
      if (acc) out_of_range = 1;
    }
  });
  if (rate == 1 && !d_raw)
    parallel_ranges(n_packed_total, [&](int64_t a0, int64_t a1) {
      const uint16_t *__restrict xs = x, *__restrict ys = y;
      const int64_t *__restrict ts = t_ns;
      uint32_t *__restrict out = xy;
      unsigned acc = 0;
      for (int64_t e = a0; e < a1; e++) {
        acc |= (unsigned)(xs[e] >= sensor_w) | (unsigned)(ys[e] >= sensor_h);
        out[e] = (uint32_t)xs[e] | ((uint32_t)ys[e] << 16) | ((uint32_t)(ts[e] < t_next_win_beg_ns) << 31);
      }
      if (acc) out_of_range = 1;
    });
  // (with sub-sampling only the sampled events were looked at: the reference reads nothing else either, but the ABI
  // promises that every event handed over is inside the sensor)
  if (!d_raw && (out_of_range.load() || rate != 1)) {
    rc = check_events(c, n, x, y, t_ns);
    if (rc) return rc;
  }
  if (err_kind.load() == CMX_ERR_TIME_ORDER)
    return fail(c, CMX_ERR_TIME_ORDER, "batch at event %lld spans a negative time interval", err_at.load());
  if (err_kind.load() == CMX_ERR_SPLINE_RANGE)
    return fail(c, CMX_ERR_SPLINE_RANGE, "batch time %lld ns outside the support of %d knots (start %lld, dt %lld)", err_at.load(), K,
                (long long)start_ns, (long long)dt_ns);
  const int nb = nbatches;
  c->order = order; c->K = K; c->num_fixed = num_fixed;
  c->batch = B; c->sample_rate = rate; c->measure = contrast_measure;
  c->knots0.resize((size_t)K);
  for (int i = 0; i < K; i++) c->knots0[i] = Quat{knots_xyzw[4 * i], knots_xyzw[4 * i + 1], knots_xyzw[4 * i + 2], knots_xyzw[4 * i + 3]};
  memset(c->h_spline, 0, sizeof(SplineArgs));
  c->h_spline->order = order;
  c->h_spline->K = K;
  c->h_spline->start_ns = start_ns;
  c->h_spline->dt_ns = dt_ns;
  blending_matrix(order, c->h_spline->blend);

  rc = ensure(c, c->d_xy, c->xy_cap, (size_t)n_packed_total);
  if (rc) return rc;
  rc = ensure(c, c->d_batch_t, c->batch_t_cap, (size_t)nb);
  if (rc) return rc;
  rc = ensure(c, c->d_poses, c->poses_cap, (size_t)nb);
  if (rc) return rc;
  rc = ensure(c, c->d_poseR, c->poseR_cap, (size_t)nb);
  if (rc) return rc;
  if (n_packed_total > 0) {
    if (d_raw)
      launch_be_pack_from_store(d_raw, reinterpret_cast<const long long *>(d_t), (long long)n, B, rate, per_batch,
                                (int)n_packed_total, (long long)t_next_win_beg_ns, c->d_xy, c->stream);
    else
      HIP_TRY(c, hipMemcpyAsync(c->d_xy, xy, (size_t)n_packed_total * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  if (nb && !d_raw) HIP_TRY(c, hipMemcpy(c->d_batch_t, bt.data(), (size_t)nb * sizeof(long long), hipMemcpyHostToDevice));
  if (nb && d_raw) {  // batch times + their validation on the device; the two error words come back with the final sync
    if (!c->d_batch_err) HIP_TRY(c, hipMalloc((void **)&c->d_batch_err, 2 * sizeof(long long)));
    long long *d_err = c->d_batch_err;
    HIP_TRY(c, hipMemsetAsync(d_err, 0, 2 * sizeof(long long), c->stream));
    launch_be_batch_times(reinterpret_cast<const long long *>(d_t), (long long)n, B, nb, (long long)start_ns, (long long)dt_ns, order,
                          K, c->d_batch_t, d_err, c->stream);
  }
  const size_t np = (size_t)c->Wp * c->Hp;
  if (IG == CMX_KEEP_MAP) {
    c->ig_nonzero = true;  // resident map: contents unknown to the host; the alpha kernel counts the non-zeros itself
  } else if (IG) {
    HIP_TRY(c, hipMemcpy(c->d_IG, IG, np * sizeof(float), hipMemcpyHostToDevice));
    std::atomic<bool> nz(false);
    parallel_ranges((int64_t)np, [&](int64_t a0, int64_t a1) {
      for (int64_t i = a0; i < a1 && !nz.load(std::memory_order_relaxed); i++)
        if (IG[i] != 0.f) nz = true;
    });
    c->ig_nonzero = nz.load();
  } else {
    HIP_TRY(c, hipMemset(c->d_IG, 0, np * sizeof(float)));
    c->ig_nonzero = false;
  }
  HIP_TRY(c, hipMemset(c->d_alpha, 0, sizeof(double)));
  c->h_result[kAlphaSlot] = 0.0;  // alpha mirror
  c->first_iter = true;     // setFirstIter(true), pose_graph_optimizer.cpp:293
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (nb && d_raw) {
    long long e[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(e, c->d_batch_err, sizeof(e), hipMemcpyDeviceToHost));
    if (e[0] == CMX_ERR_TIME_ORDER) return fail(c, CMX_ERR_TIME_ORDER, "batch at event %lld spans a negative time interval", e[1]);
    if (e[0] == CMX_ERR_SPLINE_RANGE)
      return fail(c, CMX_ERR_SPLINE_RANGE, "batch time %lld ns outside the support of %d knots (start %lld, dt %lld)", e[1], K,
                  (long long)start_ns, (long long)dt_ns);
  }
  c->n_packed = (int)n_packed_total;
  c->per_batch = per_batch;
  c->nb = nb;
  c->have_data = true;
  c->bin_valid = false;
  return CMX_OK;
}

int cmx_backend_set_window(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                           int order, int K, const double *knots_xyzw, int64_t start_ns, int64_t dt_ns,
                           int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate,
                           double blur_sigma, int contrast_measure, const float *IG) {
  return be_set_window_impl(c, n, x, y, t_ns, nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns, num_fixed,
                            t_next_win_beg_ns, event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG);
}

// ---- device-resident event store -------------------------------------------------------------------------------
static int efail(cmx_events *e, int code, const char *msg) {
  if (e) e->err = msg;
  return code;
}
int cmx_events_create(cmx_events **out, int device, int W, int H, size_t capacity) {
  if (!out) return CMX_ERR_INVALID_ARG;
  *out = nullptr;
  if (W <= 0 || H <= 0 || W > 32767 || H > 32767 || capacity == 0 || capacity > (size_t)kMaxEvents) return CMX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CMX_ERR_HIP;
  if (device < 0 || device >= ndev) return CMX_ERR_INVALID_ARG;
  cmx_events *e = new cmx_events();
  e->device = device; e->W = W; e->H = H; e->capacity = capacity;
  *out = e;
  if (hipSetDevice(device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
  for (int k = 0; k < 2; k++) {
    if (hipMalloc((void **)&e->d_xy[k], capacity * sizeof(uint32_t)) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipMalloc failed");
    if (hipMalloc((void **)&e->d_t[k], capacity * sizeof(int64_t)) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipMalloc failed");
  }
  e->h_t.reserve(capacity);
  return CMX_OK;
}
void cmx_events_destroy(cmx_events *e) {
  if (!e) return;
  hipSetDevice(e->device);
  for (int k = 0; k < 2; k++) { hipFree(e->d_xy[k]); hipFree(e->d_t[k]); }
  delete e;
}
const char *cmx_events_last_error(const cmx_events *e) { return e ? e->err.c_str() : "null event store"; }
int64_t cmx_events_begin(const cmx_events *e) { return e ? e->first_index : 0; }
int64_t cmx_events_end(const cmx_events *e) { return e ? e->first_index + (int64_t)e->size : 0; }

// append a chunk of the (time-ordered) stream: AngVelEstimator::pushEvent's events_.push_back (ang_vel_estimator.cpp:68-78)
int cmx_events_push(cmx_events *e, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns) {
  if (!e || n < 0 || (n > 0 && (!x || !y || !t_ns))) return efail(e, CMX_ERR_INVALID_ARG, "bad arguments");
  if (e->size + (size_t)n > e->capacity) return efail(e, CMX_ERR_INVALID_ARG, "event store full: drop old events first");
  if (hipSetDevice(e->device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
  std::vector<uint32_t> xy((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    if (x[i] >= e->W || y[i] >= e->H) return efail(e, CMX_ERR_EVENT_RANGE, "event coordinates outside the sensor");
    xy[(size_t)i] = (uint32_t)x[i] | ((uint32_t)y[i] << 16);
  }
  if (n) {
    if (hipMemcpy(e->d_xy[e->cur] + e->size, xy.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_t[e->cur] + e->size, t_ns, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
      return efail(e, CMX_ERR_HIP, "upload failed");
    e->h_t.insert(e->h_t.end(), t_ns, t_ns + n);
    e->size += (size_t)n;
  }
  return CMX_OK;
}

// AngVelEstimator::deleteOldEvents (ang_vel_estimator.cpp:149-173): forget everything before a global index
int cmx_events_drop_before(cmx_events *e, int64_t global_index) {
  if (!e) return CMX_ERR_INVALID_ARG;
  if (global_index <= e->first_index) return CMX_OK;
  if (global_index > e->first_index + (int64_t)e->size) return efail(e, CMX_ERR_INVALID_ARG, "index beyond the stored events");
  if (hipSetDevice(e->device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
  const size_t k = (size_t)(global_index - e->first_index), keep = e->size - k;
  const int other = 1 - e->cur;
  if (keep) {
    if (hipMemcpy(e->d_xy[other], e->d_xy[e->cur] + k, keep * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess ||
        hipMemcpy(e->d_t[other], e->d_t[e->cur] + k, keep * sizeof(int64_t), hipMemcpyDeviceToDevice) != hipSuccess)
      return efail(e, CMX_ERR_HIP, "compaction failed");
  }
  e->h_t.erase(e->h_t.begin(), e->h_t.begin() + (ptrdiff_t)k);
  e->cur = other;
  e->size = keep;
  e->first_index = global_index;
  return CMX_OK;
}

static int store_range(cmx_ctx *c, const cmx_events *e, int64_t first, int64_t count, size_t *off) {
  if (!e) return fail(c, CMX_ERR_INVALID_ARG, "null event store");
  if (!c) return CMX_ERR_INVALID_ARG;
  if (e->device != c->device || e->W != c->W || e->H != c->H)
    return fail(c, CMX_ERR_INVALID_ARG, "event store belongs to another device / sensor");
  if (count < 0 || first < e->first_index || first + count > e->first_index + (int64_t)e->size)
    return fail(c, CMX_ERR_INVALID_ARG, "range [%lld, %lld) is not held by the event store [%lld, %lld)", (long long)first,
                (long long)(first + count), (long long)e->first_index, (long long)(e->first_index + (int64_t)e->size));
  *off = (size_t)(first - e->first_index);
  return CMX_OK;
}

// packets / windows cut from the store: events_[first, first+count), exactly what getEventSubset copies
// (ang_vel_estimator.cpp:137-147, pose_graph_optimizer.cpp:131-165), without leaving the device
int cmx_frontend_set_packet_from(cmx_ctx *c, const cmx_events *e, int64_t first, int64_t count, int64_t t_ref_ns, double fx,
                                 double fy, double cx, double cy, int event_batch_size, double blur_sigma,
                                 int contrast_measure) {
  size_t off = 0;
  int rc = store_range(c, e, first, count, &off);
  if (rc) return rc;
  return fe_set_packet_impl(c, count, nullptr, nullptr, e->h_t.data() + off, e->d_xy[e->cur] + off, t_ref_ns, fx, fy, cx, cy,
                            event_batch_size, blur_sigma, contrast_measure);
}
int cmx_backend_set_window_from(cmx_ctx *c, const cmx_events *e, int64_t first, int64_t count, int order, int K,
                                const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed,
                                int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate, double blur_sigma,
                                int contrast_measure, const float *IG) {
  size_t off = 0;
  int rc = store_range(c, e, first, count, &off);
  if (rc) return rc;
  return be_set_window_impl(c, count, nullptr, nullptr, e->h_t.data() + off, e->d_xy[e->cur] + off, e->d_t[e->cur] + off, order,
                            K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size,
                            event_sample_rate, blur_sigma, contrast_measure, IG);
}

static int be_accumulate(cmx_ctx *c, const double *drotv, bool want_grad) {
  c->timing_tick++;  // see fe_accumulate
  const size_t np = (size_t)c->Wp * c->Hp;
  const int Kopt = c->K - c->num_fixed;
  c->last_adjoint = want_grad && adjoint_ok(c);
  const bool deriv = want_grad && !c->last_adjoint;
  const int P = deriv ? 3 * Kopt : 0;
  int rc = CMX_OK;
  // knot_i <- exp(drot_i) * knot_i for the non-fixed knots (CopyAndIncrementalUpdate, trajectory.cpp:240-263)
  for (int i = 0; i < c->K; i++) {
    Quat q = c->knots0[i];
    if (i >= c->num_fixed) {
      const double *d = drotv + 3 * (i - c->num_fixed);
      q = q_mul(so3_exp(d[0], d[1], d[2]), q);
    }
    c->h_spline->knots[i] = q;
  }
  {
    Span sp(c, CMX_T_POSE);
    launch_be_pose_table(*c->h_spline, c->d_batch_t, c->nb, c->order, want_grad || (adjoint_ok(c) && c->reuse_image), c->d_poseR, c->d_poses,
                         c->stream);
  }
  rc = begin_accum(c, 2 + P, np, P == 0 && adjoint_ok(c) && c->splat_mode == 1);
  if (rc) return rc;
  BeSplatArgs a = be_args(c);
  const bool use_lds = c->splat_mode == 1 && !deriv && c->n_packed > 0;
  if (use_lds && (!c->bin_valid || c->last_fallback_frac > 0.15)) {
    rc = do_binning(c, nullptr, &a);
    if (rc) return rc;
  }
  // tile occupancy: only for the LDS splat into this context's own ping-pong buffers (with a communicator attached the
  // flags are all-reduced with the planes, finish_sharded; planes owned by the caller are exchanged by the caller)
  const bool use_flags = use_lds && c->pingpong_planes > 0 && !c->accum_external;
  if (use_flags) {
    const size_t tiles = (size_t)((c->Wp + kTileX - 1) / kTileX) * ((c->Hp + kTileY - 1) / kTileY);
    if (tiles > c->tflags_cap || !c->d_tflags || !c->d_tflags_alt || !c->d_igp_flags) {
      unsigned char **ptrs[3] = {&c->d_tflags, &c->d_tflags_alt, &c->d_igp_flags};
      for (auto p : ptrs) {
        if (*p) HIP_TRY(c, hipFree(*p));
        *p = nullptr;
        HIP_TRY(c, hipMalloc((void **)p, tiles));
        HIP_TRY(c, hipMemsetAsync(*p, 0, tiles, c->stream));
      }
      c->tflags_cap = tiles;
      c->alt_flagged = false;   // whatever the partner buffer holds was written without flags
      c->igp_flags_valid = false;
    }
  }
  {
    Span sp(c, CMX_T_SPLAT, /*exact=*/true);
    c->last_used_lds = use_lds;
    if (use_lds) c->fallback_pending = true;
    if (use_lds) {
      BinnedEvents b = binned(c);
      if (use_flags) { b.tflags = c->d_tflags; b.tflags_tiles_x = (c->Wp + kTileX - 1) / kTileX; }
      launch_be_splat_lds(a, b, c->stream, sp.t0(), sp.t1());
    } else {
      launch_be_splat(a, deriv, c->stream, sp.t0(), sp.t1());
    }
  }
  c->accum_flagged = use_flags;
  HIP_TRY(c, hipGetLastError());
  c->accum_count = (size_t)(2 + P) * np;
  c->last_P = P;
  c->accumulated = true;
  c->x_valid = true;
  for (int k = 0; k < 3 * Kopt && k < 3 * kMaxKnots; k++) c->last_x[k] = drotv[k];
  return CMX_OK;
}

int cmx_backend_accumulate(cmx_ctx *c, const double *drotv, int want_grad) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_backend_set_window has not succeeded");
  if (!drotv && c->K > c->num_fixed) return fail(c, CMX_ERR_INVALID_ARG, "null drotv");
  int rc = bind(c);
  if (rc) return rc;
  return be_accumulate(c, drotv, want_grad != 0);
}

static int be_first_iter(cmx_ctx *c) {
  // first evaluation of the window: IGp <- IG, alpha <- event-density ratio (event_pano_warper.cpp:201-210)
  const size_t np = (size_t)c->Wp * c->Hp;
  if (!c->first_iter) return CMX_OK;
  if (c->ig_nonzero) {
    HIP_TRY(c, hipMemcpyAsync(c->d_IGp, c->d_IG, np * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    c->igp_flags_valid = false;
    if (c->d_igp_flags) {  // where the global map is non-zero: the image passes cannot skip those tiles
      HIP_TRY(c, hipMemsetAsync(c->d_igp_flags, 0, c->tflags_cap, c->stream));
      launch_tile_flags(c->d_IGp, c->Wp, c->Hp, c->d_igp_flags, c->stream);
      c->igp_flags_valid = true;
    }
    AlphaArgs a{};
    a.igp = c->d_IGp;
    a.il_old = c->d_accum;
    a.il_new = c->d_accum + np;
    a.npix = (int)np;
    a.nblk = 1024;
    int rc = ensure(c, c->d_partials, c->partials_cap, (size_t)5 * a.nblk);
    if (rc) return rc;
    a.partials = c->d_partials;
    a.alpha = c->d_alpha;
    a.result_alpha = c->d_result + kAlphaSlot;
    launch_alpha(a, c->stream);
    HIP_TRY(c, hipGetLastError());
  } else {
    HIP_TRY(c, hipMemsetAsync(c->d_alpha, 0, sizeof(double), c->stream));  // countNonZero(IGp) < 1 => alpha = 0
    c->h_result[kAlphaSlot] = 0.0;
  }
  c->first_iter = false;
  return CMX_OK;
}

int cmx_backend_finish(cmx_ctx *c, double *contrast, double *grad) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish without accumulate");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  const int P = 3 * (c->K - c->num_fixed);
  if (grad && c->last_P != P && !c->last_adjoint)
    return fail(c, CMX_ERR_STATE, "gradient requested but accumulate ran without it");
  int rc = bind(c);
  if (rc) return rc;
  rc = be_first_iter(c);
  if (rc) return rc;
  if (grad && c->last_adjoint) rc = run_adjoint(c, P);
  else rc = run_image_and_finalize(c, grad ? P : 0, nullptr, nullptr);
  if (rc) return rc;
  rc = sync_and_collect(c, true);
  if (rc) return rc;
  *contrast = c->h_result[0];
  if (grad) for (int k = 0; k < P; k++) grad[k] = c->h_result[2 + k];
  return CMX_OK;
}

int cmx_backend_eval(cmx_ctx *c, const double *drotv, double *contrast, double *grad) {
  const bool sharded = c && c->comm;
  if (c && c->kind == KIND_BE && drotv && can_reuse(c, drotv, 3 * (c->K - c->num_fixed), grad != nullptr)) {
    c->last_adjoint = true;
    c->reuse_hits++;
    if (sharded) return finish_sharded(c, KIND_BE, false, contrast, grad);
    return cmx_backend_finish(c, contrast, grad);
  }
  int rc = cmx_backend_accumulate(c, drotv, grad != nullptr);
  if (rc) return rc;
  if (sharded) return finish_sharded(c, KIND_BE, true, contrast, grad);
  return cmx_backend_finish(c, contrast, grad);
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
  int rc = bind(c);
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
  int rc = bind(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_size = 1;
  c->comm_rank = 0;
  return CMX_OK;
}

// ---- three-phase finish for sharded adjoint evaluations: begin (image, adjoint blur, gather -> partial gradient
// sums on the device), caller all-reduces cmx_grad_ptr(), end (finalize + read-back)
static int finish_begin(cmx_ctx *c, int kind, int want_grad) {
  if (!c || c->kind != kind) return fail(c, CMX_ERR_STATE, "wrong context kind");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish_begin without accumulate");
  int rc = bind(c);
  if (rc) return rc;
  const int P = (kind == KIND_FE) ? 3 : 3 * (c->K - c->num_fixed);
  if (kind == KIND_BE) {
    rc = be_first_iter(c);
    if (rc) return rc;
  }
  if (want_grad && c->last_adjoint) {
    rc = run_adjoint(c, P, 1);
    c->pending_P = P;
  } else {
    if (want_grad && c->last_P != P) return fail(c, CMX_ERR_STATE, "gradient requested but accumulate ran without it");
    rc = run_image_and_finalize(c, want_grad ? P : 0, nullptr, nullptr);
    c->pending_P = -1;  // nothing left to exchange: finish_end only synchronises
  }
  if (rc) return rc;
  c->finish_pending = true;
  return CMX_OK;
}
static int finish_end(cmx_ctx *c, int kind, double *contrast, double *grad) {
  if (!c || c->kind != kind) return fail(c, CMX_ERR_STATE, "wrong context kind");
  if (!c->finish_pending) return fail(c, CMX_ERR_STATE, "finish_end without finish_begin");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  int rc = bind(c);
  if (rc) return rc;
  const int P = (kind == KIND_FE) ? 3 : 3 * (c->K - c->num_fixed);
  if (c->pending_P >= 0) {
    rc = run_adjoint(c, c->pending_P, 2);
    if (rc) return rc;
  }
  c->finish_pending = false;
  rc = sync_and_collect(c, true);
  if (rc) return rc;
  *contrast = c->h_result[0];
  if (grad) for (int k = 0; k < P; k++) grad[k] = c->h_result[2 + k];
  return CMX_OK;
}
int cmx_frontend_finish_begin(cmx_ctx *c, int want_grad) { return finish_begin(c, KIND_FE, want_grad); }
int cmx_frontend_finish_end(cmx_ctx *c, double *contrast, double *grad) { return finish_end(c, KIND_FE, contrast, grad); }
int cmx_backend_finish_begin(cmx_ctx *c, int want_grad) { return finish_begin(c, KIND_BE, want_grad); }
int cmx_backend_finish_end(cmx_ctx *c, double *contrast, double *grad) { return finish_end(c, KIND_BE, contrast, grad); }
void *cmx_grad_ptr(const cmx_ctx *c) { return c ? c->d_gsum : nullptr; }
size_t cmx_grad_count(const cmx_ctx *c) { return (c && c->finish_pending && c->pending_P > 0) ? (size_t)(2 * c->pending_P) : 0; }
int cmx_set_grad_buffer(cmx_ctx *c, void *device_ptr, size_t n_doubles) {
  if (!c) return CMX_ERR_INVALID_ARG;
  int rc = bind(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!c->gsum_external && c->d_gsum) HIP_TRY(c, hipFree(c->d_gsum));
  c->d_gsum = (double *)device_ptr;
  c->gsum_cap = device_ptr ? n_doubles : 0;
  c->gsum_external = device_ptr != nullptr;
  return CMX_OK;
}

// ---- global-map upkeep on the device: IG and the visit counts stay resident
int cmx_backend_update_map(cmx_ctx *c, int max_update_times) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "no evaluation has run in this window (IL_old undefined)");
  int rc = bind(c);
  if (rc) return rc;
  launch_update_map(c->d_IG, c->d_accum, c->d_visits, c->Wp * c->Hp, max_update_times, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMX_OK;
}
int cmx_backend_mark_visited(cmx_ctx *c, const double q[4], int radius) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!q || radius < 0 || radius > 64) return fail(c, CMX_ERR_INVALID_ARG, "bad pose / radius");
  int rc = bind(c);
  if (rc) return rc;
  const Mat3 R = q_to_R(Quat{q[0], q[1], q[2], q[3]});
  launch_mark_visited(be_args(c), R.m, c->H, radius, c->d_mask, c->d_visits, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMX_OK;
}
int cmx_backend_reset_map(cmx_ctx *c) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind(c);
  if (rc) return rc;
  const size_t np = (size_t)c->Wp * c->Hp;
  HIP_TRY(c, hipMemsetAsync(c->d_IG, 0, np * sizeof(float), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_visits, 0, np, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->ig_nonzero = false;
  return CMX_OK;
}
int cmx_backend_get_map(cmx_ctx *c, float *IG, unsigned char *visits) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind(c);
  if (rc) return rc;
  const size_t np = (size_t)c->Wp * c->Hp;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (IG) HIP_TRY(c, hipMemcpy(IG, c->d_IG, np * sizeof(float), hipMemcpyDeviceToHost));
  if (visits) HIP_TRY(c, hipMemcpy(visits, c->d_visits, np, hipMemcpyDeviceToHost));
  return CMX_OK;
}
int cmx_backend_set_map(cmx_ctx *c, const float *IG, const unsigned char *visits) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind(c);
  if (rc) return rc;
  const size_t np = (size_t)c->Wp * c->Hp;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (IG) HIP_TRY(c, hipMemcpy(c->d_IG, IG, np * sizeof(float), hipMemcpyHostToDevice));
  if (visits) HIP_TRY(c, hipMemcpy(c->d_visits, visits, np, hipMemcpyHostToDevice));
  return CMX_OK;
}

int cmx_backend_get_plane(cmx_ctx *c, int which, float *host) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "no evaluation has run in this window");
  if (!host) return fail(c, CMX_ERR_INVALID_ARG, "null host buffer");
  int rc = bind(c);
  if (rc) return rc;
  const size_t np = (size_t)c->Wp * c->Hp;
  const float *src = nullptr;
  if (which == CMX_PLANE_IL_OLD) src = c->d_accum;
  else if (which == CMX_PLANE_IL_NEW) src = c->d_accum + np;
  else if (which == CMX_PLANE_IWE || (which >= CMX_PLANE_DERIV0 && which < CMX_PLANE_DERIV0 + c->last_P)) {
    const int P = (which == CMX_PLANE_IWE) ? 0 : c->last_P;
    rc = ensure(c, c->d_scratch, c->scratch_cap, (size_t)(1 + P) * np);
    if (rc) return rc;
    rc = be_first_iter(c);
    if (rc) return rc;
    rc = run_image_and_finalize(c, P, c->d_scratch, P ? c->d_scratch + np : nullptr);
    if (rc) return rc;
    src = (which == CMX_PLANE_IWE) ? c->d_scratch : c->d_scratch + (size_t)(1 + which - CMX_PLANE_DERIV0) * np;
  } else {
    return fail(c, CMX_ERR_INVALID_ARG, "plane %d not available", which);
  }
  HIP_TRY(c, hipMemcpyAsync(host, src, np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  return sync_and_collect(c);
}

int cmx_backend_get_alpha(cmx_ctx *c, double *alpha) {
  if (!c || c->kind != KIND_BE || !alpha) return fail(c, CMX_ERR_INVALID_ARG, "bad argument");
  int rc = bind(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *alpha = c->h_result[kAlphaSlot];
  return CMX_OK;
}

}  // extern "C"
