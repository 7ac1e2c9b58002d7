// cmx_context.cpp -- context life cycle, options, timing and error text of the C ABI (include/cmax_hip.h), plus the
// host-side helpers every entry point shares.  All compute is in cmx_kernels.hip / cmx_binning.hip.
#include <condition_variable>
#include <mutex>

#include "cmx_context.hpp"

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

int bind_device(cmx_ctx *c) {
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
  c->Mx_radius = -1;
  int built_axes = 0;  // bit per axis whose banded operator was rebuilt for THIS radius
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
    if (r >= 1 && r <= kMaxRadius && L > 4 * r) {  // (both ends: image_adjoint2 / 2g)
      // banded composite operator M = G^T G of this axis: (M x)[q] = sum_i M[q][i] x[q - 2r + i], where
      // G[p][s] = sum_j taps[r+j] [reflect101(p+j) == s] is the REFLECT_101 blur (the forward pass of the image kernels)
      const int bw = 4 * r + 1;
      std::vector<double> G((size_t)L * (2 * r + 1), 0.0);  // G[p][s - (p - r)] for s in [p-r, p+r]
      auto refl = [L](int p) { if (L == 1) return 0; while (p < 0 || p >= L) p = p < 0 ? -p : 2 * (L - 1) - p; return p; };
      for (int p = 0; p < L; p++)
        for (int j = -r; j <= r; j++) {
          const int s2 = refl(p + j);
          if (s2 >= p - r && s2 <= p + r) G[(size_t)p * (2 * r + 1) + (s2 - (p - r))] += (double)c->taps[r + j];
        }
      std::vector<float> M((size_t)L * bw, 0.f);
      for (int q = 0; q < L; q++)
        for (int i = 0; i < bw; i++) {
          const int s2 = q - 2 * r + i;
          if (s2 < 0 || s2 >= L) continue;
          double acc = 0;
          for (int p = std::max(0, std::max(q, s2) - r); p <= std::min(L - 1, std::min(q, s2) + r); p++)
            acc += G[(size_t)p * (2 * r + 1) + (q - (p - r))] * G[(size_t)p * (2 * r + 1) + (s2 - (p - r))];
          M[(size_t)q * bw + i] = (float)acc;
        }
      float *&dm = axis == 0 ? c->d_Mx : c->d_My;
      size_t &mcap = axis == 0 ? c->Mx_cap : c->My_cap;
      rc = ensure(c, dm, mcap, M.size());
      if (rc) return rc;
      HIP_TRY(c, hipMemcpy(dm, M.data(), M.size() * sizeof(float), hipMemcpyHostToDevice));
      built_axes |= 1 << axis;
    }
  }
  // the image kernels take the composite form only with BOTH tables of this radius: a tiny image (one side <= 4r) keeps
  // the general passes instead of pairing a fresh table with a stale one of an earlier sigma
  if (built_axes == 3) c->Mx_radius = r;
  return CMX_OK;
}

// cv::GaussianBlur(Size(0,0), sigma) on CV_32F: ksize = cvRound(sigma*8+1)|1; fp64 kernel normalised, cast to fp32
int setup_blur(cmx_ctx *c, double sigma) {
  // same sigma as last time (the image size of a context never changes): taps, G^T 1 factors and operator tables are on the
  // device already -- a packet / window no longer pays four to six synchronous table uploads
  if (c->blur_sigma_built == sigma && sigma >= 0) return CMX_OK;
  c->blur_sigma_built = -1.0;
  c->sigma = sigma;
  if (!(sigma > 0)) {
    c->radius = 0;
    c->taps[0] = 1.f;
    const int rc0 = upload_gt1(c);
    if (!rc0) c->blur_sigma_built = sigma;
    return rc0;
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
  const int rc1 = upload_gt1(c);
  if (!rc1) c->blur_sigma_built = sigma;
  return rc1;
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

int create_common(cmx_ctx **out, int kind, int device, int W, int H, const double *lut) {
  if (!out) return CMX_ERR_INVALID_ARG;
  *out = nullptr;
  if (W <= 0 || H <= 0 || W > 32767 || H > 32767 || !lut) return CMX_ERR_INVALID_ARG;
  if ((long long)W * H > kMaxPixels) return CMX_ERR_INVALID_ARG;  // kernels index pixels with 32-bit ints
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
  HIP_TRY(c, hipHostMalloc((void **)&c->h_result2, c->result_cap * sizeof(double), hipHostMallocMapped));
  HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_result2, c->h_result2, 0));
  memset(c->h_result2, 0, c->result_cap * sizeof(double));
  HIP_TRY(c, hipMalloc((void **)&c->d_gate, sizeof(int)));
  HIP_TRY(c, hipMemset(c->d_gate, 0, sizeof(int)));
  HIP_TRY(c, hipMalloc((void **)&c->d_tail_counters, kTailCounterWords * sizeof(unsigned)));
  HIP_TRY(c, hipMemset(c->d_tail_counters, 0, kTailCounterWords * sizeof(unsigned)));
  HIP_TRY(c, hipMalloc((void **)&c->d_gacc, (size_t)kTailShards * kGaccStride * sizeof(double)));
  HIP_TRY(c, hipMemset(c->d_gacc, 0, (size_t)kTailShards * kGaccStride * sizeof(double)));
  HIP_TRY(c, hipDeviceSynchronize());  // null-stream clears / copies above vs the context's non-blocking stream
  return CMX_OK;
}

int ensure_accum(cmx_ctx *c, size_t need);

int check_event_args(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t) {
  if (n < 0 || n > kMaxEvents) return fail(c, CMX_ERR_INVALID_ARG, "bad event count %lld (limit %lld)", (long long)n, (long long)kMaxEvents);
  if (n > 0 && (!x || !y || !t)) return fail(c, CMX_ERR_INVALID_ARG, "null event arrays");
  return CMX_OK;
}
int make_aos(cmx_ctx *c, int64_t n, const void *events, const cmx_aos_layout *layout, EvAos *out) {
  if (n < 0 || n > kMaxEvents) return fail(c, CMX_ERR_INVALID_ARG, "bad event count %lld (limit %lld)", (long long)n, (long long)kMaxEvents);
  if (!layout || (n > 0 && !events)) return fail(c, CMX_ERR_INVALID_ARG, "null event array / layout");
  const size_t st = layout->stride;
  if (st < 12 || layout->off_x + 2 > st || layout->off_y + 2 > st || layout->off_sec + 4 > st || layout->off_nsec + 4 > st)
    return fail(c, CMX_ERR_INVALID_ARG, "record layout: fields outside the %zu-byte record", st);
  out->base = static_cast<const unsigned char *>(events);
  out->stride = st; out->ox = layout->off_x; out->oy = layout->off_y; out->os = layout->off_sec; out->on = layout->off_nsec;
  return CMX_OK;
}
int check_events(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t, const EvAos *aos) {
  if (!aos) {
    int rc0 = check_event_args(c, n, x, y, t);
    if (rc0) return rc0;
  }
  const int W = c->W, H = c->H;
  std::atomic<int64_t> bad(-1);
  auto X = [&](int64_t i) { return aos ? aos->X(i) : (unsigned)x[i]; };
  auto Y = [&](int64_t i) { return aos ? aos->Y(i) : (unsigned)y[i]; };
  parallel_ranges(n, [&](int64_t a, int64_t b) {
    unsigned acc = 0;
    for (int64_t i = a; i < b; i++) acc |= (unsigned)(X(i) >= (unsigned)W) | (unsigned)(Y(i) >= (unsigned)H);
    if (acc)
      for (int64_t i = a; i < b; i++)
        if (X(i) >= (unsigned)W || Y(i) >= (unsigned)H) {
          int64_t cur = bad.load();
          while ((cur < 0 || i < cur) && !bad.compare_exchange_weak(cur, i)) {}
          break;
        }
  });
  const int64_t i = bad.load();
  if (i >= 0)
    return fail(c, CMX_ERR_EVENT_RANGE, "event %lld at (%u,%u) outside the %dx%d sensor", (long long)i, X(i), Y(i), W, H);
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

int ensure_pinned_dts(cmx_ctx *c, size_t n) {
  if (n <= c->h_dts_cap && c->h_dts) return CMX_OK;
  if (c->h_dts) HIP_TRY(c, hipHostFree(c->h_dts));
  c->h_dts = nullptr;
  c->h_dts_cap = 0;
  const size_t cap = n + n / 4 + 256;
  HIP_TRY(c, hipHostMalloc((void **)&c->h_dts, cap * sizeof(double), hipHostMallocDefault));
  c->h_dts_cap = cap;
  return CMX_OK;
}

// =============================================================================================== generic
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
  if (c->group) {  // a group's handle: workers, transport and every member (which come back here with group == nullptr)
    if (c->group_rank == 0) group_destroy(c);
    return;
  }
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto &s : c->spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
  for (auto e : c->event_pool) hipEventDestroy(e);
  hipFree(c->d_lut);
  hipFree(c->d_lut2);
  hipFree(c->d_nchunks);
  if (c->h_nchunks) hipHostFree(c->h_nchunks);
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
  hipFree(c->d_hist);
  hipFree(c->d_sb);
  hipFree(c->d_tb);
  hipFree(c->d_sdt);
  hipFree(c->d_fixed);
  hipFree(c->d_tile_start);
  hipFree(c->d_chunks);
  hipFree(c->d_fallback);
  hipFree(c->d_fnbr_expected); hipFree(c->d_fnbr_cnt); hipFree(c->d_fpartials); hipFree(c->d_fuse_trace);
  hipFree(c->d_ftile_done); hipFree(c->d_ftiles_done); hipFree(c->d_fn_active);
  hipFree(c->d_itilde);
  hipFree(c->d_cx);
  hipFree(c->d_cy);
  hipFree(c->d_Mx);
  hipFree(c->d_My);
  hipFree(c->d_gpartials);
  hipFree(c->d_tflags); hipFree(c->d_tflags_alt); hipFree(c->d_igp_flags);
  hipFree(c->d_xlist[0]); hipFree(c->d_xlist[1]); hipFree(c->d_xmember[0]); hipFree(c->d_xmember[1]); hipFree(c->d_xmiss); hipFree(c->d_xstage); hipFree(c->d_xstage_b); hipFree(c->d_xstage_out);
  hipFree(c->d_tile_list); hipFree(c->d_tile_count);
  hipFree(c->d_vparts);
  hipFree(c->d_tail_counters);
  hipFree(c->d_gacc);
  if (!c->gsum_external) hipFree(c->d_gsum);
  if (c->h_result) hipHostFree(c->h_result);
  if (c->h_result2) hipHostFree(c->h_result2);
  if (c->h_dts) hipHostFree(c->h_dts);
  hipFree(c->d_gate);
  hipFree(c->d_chain);
  if (c->h_chain_ring) hipHostFree(c->h_chain_ring);
  if (c->h_chain_init) hipHostFree(c->h_chain_init);
  if (c->h_many) hipHostFree(c->h_many);
  comm_release(c);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

static int set_option_one(cmx_ctx *c, int key, int value);
int cmx_set_option(cmx_ctx *c, int key, int value) {
  if (!c) return CMX_ERR_INVALID_ARG;
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return set_option_one(m, key, value); });  // members must agree
  return set_option_one(c, key, value);
}
static int set_option_one(cmx_ctx *c, int key, int value) {
  switch (key) {
    case CMX_OPT_GRAD_MODE:
      if (value != CMX_GRAD_PLANES && value != CMX_GRAD_ADJOINT) return fail(c, CMX_ERR_INVALID_ARG, "bad grad mode %d", value);
      c->grad_mode = value;
      c->x_valid = false;  // the resident pose table may have been built without Jacobians: never reuse across a mode switch
      return CMX_OK;
    case CMX_OPT_SPLAT_MODE:
      if (value != 0 && value != 1) return fail(c, CMX_ERR_INVALID_ARG, "bad splat mode %d", value);
      c->splat_mode = value;
      c->bin_valid = false;
      c->x_valid = false;
      return CMX_OK;
    case CMX_OPT_REUSE_IMAGE:
      c->reuse_image = value != 0;
      c->x_valid = false;  // (the pose table of a cost-only evaluation carries Jacobians only when reuse was on)
      return CMX_OK;
    case CMX_OPT_DETERMINISTIC:
      c->deterministic = value != 0;
      c->x_valid = false;  // a resident image of the other mode is not reused
      return CMX_OK;
    case CMX_OPT_SPIN_WAIT:
      if (value < 0 || value > 1000000) return fail(c, CMX_ERR_INVALID_ARG, "bad spin budget %d", value);
      c->ticket_wait = value != 0;
      c->spin_eval_us = value == 1 ? -1 : value;  // 1: an evaluation is waited for on its ticket however long it runs
      c->spin_idle_us = value == 1 ? 50 : value;  // threads with nothing on the device give their core back after 50 us by default
      return CMX_OK;
    case CMX_OPT_TAIL_FINALIZE:
      c->tail_poll = value == 3;  // 3: the tail as 1, with the polling form on the front-end gather (A/B: measured, no gain)
      c->tail_finalize = value == 3 ? 1 : (value < 0 ? 0 : (value > 2 ? 2 : value));
      return CMX_OK;
    case CMX_OPT_GATED_DF:
      c->gated_df = value != 0;
      c->gated_pending = false;
      return CMX_OK;
    case CMX_OPT_FOLD_BATCH:
      c->fold_batch = value != 0;
      return CMX_OK;
    case CMX_OPT_CHAIN_SOLVE:
      c->chain_solve = value != 0;
      c->chain_test = value == 2 ? 1 : (value == 3 ? 2 : 0);
      c->chain_self_gating = value != 4;
      return CMX_OK;
    case CMX_OPT_FUSED_IMAGE:
      c->fused_image = value != 0;
      c->fused_full = value == 2;  // 2: gather + finalize inside the same launch as well (A/B: measured slower, see cmax_hip_diag.h)
      c->fused_self = value == 3;  // 3: one launch of the chunk workgroups alone (cmx_selfserve.hpp; A/B: measured a tie with 1)
      c->fused_self_strikes = 0;
      c->bin_valid = false;  // the fused pass's tables are built with the chunk table
      c->x_valid = false;
      return CMX_OK;
    case CMX_OPT_COMPOSITE_IMAGE:
      c->composite_image = value != 0;
      c->x_valid = false;  // a resident Jt of the other form is not reused
      return CMX_OK;
    default: return fail(c, CMX_ERR_INVALID_ARG, "unknown option %d", key);
  }
}

int cmx_hint_next_df(cmx_ctx *c, double threshold, int mode) {
  if (!c || mode < 0 || mode > 4) return CMX_ERR_INVALID_ARG;
  c->gate_thr = threshold;
  c->gate_mode = mode;
  return CMX_OK;
}

int cmx_set_stream(cmx_ctx *c, void *hip_stream) {
  if (!c) return CMX_ERR_INVALID_ARG;
  CMX_NOT_FOR_GROUPS(c, "a caller-owned stream");
  int rc = bind_device(c);
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

// ---- the context's own stream with a scheduling priority or a compute-unit mask: a front-end context beside a back-end
// window solve on one GPU (the reference's two threads, src/node.cpp:22 + src/cmax_slam.cpp:92)
static int replace_own_stream(cmx_ctx *c, int priority_level, const uint32_t *mask, int n_words) {
  int rc = bind_device(c);
  if (rc) return rc;
  if (c->stream && !c->own_stream) return fail(c, CMX_ERR_STATE, "the context runs on a caller-owned stream (cmx_set_stream)");
  if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
  hipStream_t s = nullptr;
  if (mask && n_words > 0) {
    HIP_TRY(c, hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
  } else {
    int least = 0, greatest = 0;  // numerically: greatest priority <= least priority (HIP: lower number = served first)
    HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = priority_level > 0 ? greatest : (priority_level < 0 ? least : (least + greatest) / 2);
    HIP_TRY(c, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  }
  if (c->stream) HIP_TRY(c, hipStreamDestroy(c->stream));
  c->stream = s;
  c->own_stream = true;
  return CMX_OK;
}
int cmx_set_stream_priority(cmx_ctx *c, int level) {
  if (!c) return CMX_ERR_INVALID_ARG;
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return replace_own_stream(m, level, nullptr, 0); });
  return replace_own_stream(c, level, nullptr, 0);
}
int cmx_set_cu_mask(cmx_ctx *c, const uint32_t *mask, int n_words) {
  if (!c || n_words < 0 || (n_words > 0 && !mask)) return CMX_ERR_INVALID_ARG;
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return replace_own_stream(m, 0, mask, n_words); });
  return replace_own_stream(c, 0, mask, n_words);
}

// ---- cooperative scheduling (see cmx_set_sched_class in the header).  Stream priorities do not pre-empt the back end's resident
// workgroups and CU masks split the chip statically (profiles/r04_fe_beside_be.txt); what the two paths of the reference need is
// that the back end does not START an evaluation while the front end's short solve is on the device.
namespace {
constexpr int kSchedDevices = 64;
constexpr long long kUrgentLingerNs = 20000, kYieldCapNs = 5000000;
std::atomic<int> g_urgent_active[kSchedDevices];
std::atomic<long long> g_urgent_last_end_ns[kSchedDevices];
// a background context that has spun for its idle budget sleeps here until the urgent burst ends (or its starvation guard expires)
std::mutex g_urgent_mu[kSchedDevices];
std::condition_variable g_urgent_cv[kSchedDevices];
std::atomic<int> g_urgent_sleepers[kSchedDevices];
long long sched_now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace
UrgentScope::UrgentScope(cmx_ctx *ctx) : c((ctx && ctx->sched_class > 0 && ctx->device >= 0 && ctx->device < kSchedDevices) ? ctx : nullptr) {
  if (c) g_urgent_active[c->device].fetch_add(1, std::memory_order_acq_rel);
}
UrgentScope::~UrgentScope() {
  if (!c) return;
  const int d = c->device;
  g_urgent_last_end_ns[d].store(sched_now_ns(), std::memory_order_release);
  if (g_urgent_active[d].fetch_sub(1, std::memory_order_seq_cst) == 1 && g_urgent_sleepers[d].load(std::memory_order_seq_cst) > 0) {
    std::lock_guard<std::mutex> lk(g_urgent_mu[d]);
    g_urgent_cv[d].notify_all();
  }
}
void yield_to_urgent(cmx_ctx *c) {
  if (!c || c->sched_class >= 0 || c->device < 0 || c->device >= kSchedDevices) return;
  const int d = c->device;
  const long long spin_ns = (long long)c->spin_idle_us * 1000;
  long long t0 = 0;
  for (unsigned spins = 0;; spins++) {
    if (g_urgent_active[d].load(std::memory_order_acquire) == 0) {
      const long long last = g_urgent_last_end_ns[d].load(std::memory_order_acquire);
      if (last == 0) return;  // no urgent context has ever run on this device
      const long long now = sched_now_ns();
      if (now - last >= kUrgentLingerNs) return;  // (the linger is 20 us: spun through, never slept)
      if (!t0) t0 = now;
      if (now - t0 > kYieldCapNs) return;
    } else if ((spins & 63u) == 63u || spin_ns == 0) {
      const long long now = sched_now_ns();
      if (!t0) t0 = now;
      if (now - t0 > kYieldCapNs) return;  // never starve: an urgent caller that stays busy gets at most this much in a row
      if (now - t0 >= spin_ns) {
        // the burst outlasts the spin budget (a front-end solve is ~0.5 ms): give the core back until it ends
        std::unique_lock<std::mutex> lk(g_urgent_mu[d]);
        g_urgent_sleepers[d].fetch_add(1, std::memory_order_seq_cst);
        g_urgent_cv[d].wait_for(lk, std::chrono::nanoseconds(kYieldCapNs - (now - t0)),
                                [&] { return g_urgent_active[d].load(std::memory_order_seq_cst) == 0; });
        g_urgent_sleepers[d].fetch_sub(1, std::memory_order_seq_cst);
        continue;
      }
    }
    if (!t0) t0 = sched_now_ns();
    __builtin_ia32_pause();
  }
}
int cmx_set_sched_class(cmx_ctx *c, int sched_class) {
  if (!c) return CMX_ERR_INVALID_ARG;
  if (sched_class < CMX_SCHED_BACKGROUND || sched_class > CMX_SCHED_URGENT) return fail(c, CMX_ERR_INVALID_ARG, "bad scheduling class %d", sched_class);
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { m->sched_class = sched_class; return (int)CMX_OK; });
  c->sched_class = sched_class;
  return CMX_OK;
}

int cmx_abi_version(void) { return CMX_ABI_VERSION; }

int cmx_get_stats(cmx_ctx *c, double *out, int n_stats) {
  if (!c || !out || n_stats < 0) return CMX_ERR_INVALID_ARG;
  double stats[CMX_N_STATS];
  for (int i = 0; i < CMX_N_STATS; i++) stats[i] = 0;
  stats[0] = (double)c->rebin_count;
  stats[1] = c->last_fallback_frac;
  {  // true length of the chunk table (device-resident until the first evaluation after a binning has been collected)
    int nch = c->nchunks;
    if (!c->nchunks_exact && c->d_nchunks && c->bin_valid && bind_device(c) == CMX_OK && hipStreamSynchronize(c->stream) == hipSuccess)
      (void)hipMemcpy(&nch, c->d_nchunks, sizeof(int), hipMemcpyDeviceToHost);
    stats[2] = (double)nch;
  }
  stats[3] = (double)c->n_packed;
  stats[4] = (double)c->reuse_hits;
  stats[5] = (double)c->sharded_host_syncs;
  stats[6] = (double)c->xset_misses;
  stats[7] = (double)c->xset_n;
  stats[8] = (double)c->comm_bytes_eval;
  stats[9] = (double)c->spec_images;
  stats[10] = (double)c->spec_hits;
  stats[11] = (double)c->gated_launches;
  stats[12] = (double)c->gated_hits;
  stats[13] = (double)c->chain_solves;
  stats[14] = (double)c->chain_slots;
  stats[15] = (double)c->chain_takeovers;
  stats[16] = (double)c->chain_warm_starts;
  stats[CMX_STAT_FUSED_EVALS] = (double)c->fused_evals;
  stats[CMX_STAT_FUSED_REDOS] = (double)c->fused_redos;
  stats[CMX_STAT_ONE_LAUNCH_EVALS] = (double)c->fused_full_evals;
  stats[CMX_STAT_FUSED_TIMEOUTS] = (double)c->fused_timeouts;
  stats[CMX_STAT_SELF_SERVE_EVALS] = (double)c->fused_self_evals;
  for (int i = 0; i < n_stats && i < CMX_N_STATS; i++) out[i] = stats[i];
  return CMX_OK;
}

int cmx_timing_enable(cmx_ctx *c, int on) {
  if (!c) return CMX_ERR_INVALID_ARG;
  c->timing = (on & 0xff) != 0;
  c->timing_mask = on & 0xff;  // CMX_T_COUNT <= 8 classes
  c->timing_every = (on >> 8) > 0 ? (on >> 8) : 1;  // bits 8..: sample every n-th evaluation only
  c->timing_tick = 0;
  return CMX_OK;
}
int cmx_timing_get(cmx_ctx *c, double ms[CMX_T_COUNT], int64_t launches[CMX_T_COUNT]) {
  if (!c) return CMX_ERR_INVALID_ARG;
  if (!c->spans.empty()) {
    int rc = bind_device(c);
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
  CMX_NOT_FOR_GROUPS(c, "a caller-owned accumulation buffer");
  int rc = bind_device(c);
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

