// cmx_events.cpp -- cmx_events_*: the device-resident event store and the packets /
// windows cut from it (AngVelEstimator::pushEvent / getEventSubset, PoseGraphOptimizer::getEventSubset).
#include "cmx_context.hpp"

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
    // a device-to-device hipMemcpy may return before the copy has run, and the contexts that cut packets / windows from
    // the store use non-blocking streams: make the compaction complete before the buffers are flipped
    if (hipDeviceSynchronize() != hipSuccess) return efail(e, CMX_ERR_HIP, "compaction failed");
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
  CMX_NOT_FOR_GROUPS(c, "a window cut from a device event store (the store lives on one device)");
  size_t off = 0;
  int rc = store_range(c, e, first, count, &off);
  if (rc) return rc;
  return be_set_window_impl(c, count, nullptr, nullptr, e->h_t.data() + off, e->d_xy[e->cur] + off, e->d_t[e->cur] + off, order,
                            K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size,
                            event_sample_rate, blur_sigma, contrast_measure, IG);
}

