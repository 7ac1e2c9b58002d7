// cmx_events.cpp -- cmx_events_*: the device-resident event store and the packets /
// windows cut from it (AngVelEstimator::pushEvent / getEventSubset, PoseGraphOptimizer::getEventSubset).
#include "cmx_context.hpp"

// ---- device-resident event store -------------------------------------------------------------------------------
static int efail(cmx_events *e, int code, const char *msg) {
  if (e) e->err = msg;
  return code;
}
static int events_create(cmx_events **out, const int *devices, int n_devices, int W, int H, size_t capacity) {
  if (!out) return CMX_ERR_INVALID_ARG;
  *out = nullptr;
  if (!devices || n_devices < 1 || W <= 0 || H <= 0 || W > 32767 || H > 32767 || capacity == 0 || capacity > (size_t)kMaxEvents)
    return CMX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CMX_ERR_HIP;
  for (int k = 0; k < n_devices; k++)
    if (devices[k] < 0 || devices[k] >= ndev) return CMX_ERR_INVALID_ARG;
  cmx_events *e = new cmx_events();
  e->device = devices[0]; e->W = W; e->H = H; e->capacity = capacity;
  *out = e;
  for (int k = 0; k < n_devices; k++) {
    if (e->on(devices[k])) continue;  // members that share a device share its replica
    cmx_events::Replica r;
    r.device = devices[k];
    if (hipSetDevice(r.device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
    e->rep.push_back(r);  // (pushed first: cmx_events_destroy frees whatever a failed allocation left behind)
    cmx_events::Replica &q = e->rep.back();
    if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipStreamCreate failed");
    for (int b = 0; b < 2; b++) {
      if (hipMalloc((void **)&q.d_xy[b], capacity * sizeof(uint32_t)) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipMalloc failed");
      if (hipMalloc((void **)&q.d_t[b], capacity * sizeof(int64_t)) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipMalloc failed");
    }
  }
  e->h_t.reserve(capacity);
  (void)hipSetDevice(e->device);
  return CMX_OK;
}
int cmx_events_create(cmx_events **out, int device, int W, int H, size_t capacity) {
  return events_create(out, &device, 1, W, H, capacity);
}
int cmx_events_create_group(cmx_events **out, const int *devices, int n_devices, int W, int H, size_t capacity) {
  return events_create(out, devices, n_devices, W, H, capacity);
}
void cmx_events_destroy(cmx_events *e) {
  if (!e) return;
  for (cmx_events::Replica &r : e->rep) {
    (void)hipSetDevice(r.device);
    if (r.stream) { (void)hipStreamSynchronize(r.stream); (void)hipStreamDestroy(r.stream); }
    for (int k = 0; k < 2; k++) { (void)hipFree(r.d_xy[k]); (void)hipFree(r.d_t[k]); }
  }
  if (e->h_xy) (void)hipHostFree(e->h_xy);
  if (e->h_tp) (void)hipHostFree(e->h_tp);
  delete e;
}
const char *cmx_events_last_error(const cmx_events *e) { return e ? e->err.c_str() : "null event store"; }
int64_t cmx_events_begin(const cmx_events *e) { return e ? e->first_index : 0; }
int64_t cmx_events_end(const cmx_events *e) { return e ? e->first_index + (int64_t)e->size : 0; }
int cmx_events_devices(const cmx_events *e, int *devices, int max_devices) {
  if (!e) return 0;
  for (size_t k = 0; k < e->rep.size() && (int)k < max_devices && devices; k++) devices[k] = e->rep[k].device;
  return (int)e->rep.size();
}

// append a chunk of the (time-ordered) stream: AngVelEstimator::pushEvent's events_.push_back (ang_vel_estimator.cpp:68-78).
// One packing pass on the host into pinned staging, then one asynchronous upload per replica (the devices copy side by side).
static int events_push_impl(cmx_events *e, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns, const EvAos *aos);
int cmx_events_push(cmx_events *e, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns) {
  if (!e || n < 0 || (n > 0 && (!x || !y || !t_ns))) return efail(e, CMX_ERR_INVALID_ARG, "bad arguments");
  return events_push_impl(e, n, x, y, t_ns, nullptr);
}
// the same from the host's own records (dvs_msgs::Event, cmx_aos_layout): pushEvent's loop over msg->events
// (ang_vel_estimator.cpp:68-78) as ONE call -- x | y << 16 and sec * 1e9 + nsec are formed in the packing pass
int cmx_events_push_aos(cmx_events *e, int64_t n, const void *events, const cmx_aos_layout *layout) {
  if (!e || n < 0 || !layout || (n > 0 && !events)) return efail(e, CMX_ERR_INVALID_ARG, "bad arguments");
  const size_t st = layout->stride;
  if (st < 12 || layout->off_x + 2 > st || layout->off_y + 2 > st || layout->off_sec + 4 > st || layout->off_nsec + 4 > st)
    return efail(e, CMX_ERR_INVALID_ARG, "record layout: fields outside the record");
  EvAos aos;
  aos.base = static_cast<const unsigned char *>(events);
  aos.stride = st; aos.ox = layout->off_x; aos.oy = layout->off_y; aos.os = layout->off_sec; aos.on = layout->off_nsec;
  return events_push_impl(e, n, nullptr, nullptr, nullptr, &aos);
}
static int events_push_impl(cmx_events *e, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns, const EvAos *aos) {
  if (e->size + (size_t)n > e->capacity) return efail(e, CMX_ERR_INVALID_ARG, "event store full: drop old events first");
  if (n == 0) return CMX_OK;
  if ((size_t)n > e->stage_cap) {
    if (e->h_xy) (void)hipHostFree(e->h_xy);
    if (e->h_tp) (void)hipHostFree(e->h_tp);
    e->h_xy = nullptr; e->h_tp = nullptr; e->stage_cap = 0;
    const size_t cap = std::max<size_t>((size_t)n, 1u << 16);
    if (hipHostMalloc((void **)&e->h_xy, cap * sizeof(uint32_t), hipHostMallocPortable) != hipSuccess ||
        hipHostMalloc((void **)&e->h_tp, cap * sizeof(int64_t), hipHostMallocPortable) != hipSuccess)
      return efail(e, CMX_ERR_HIP, "pinned staging allocation failed");
    e->stage_cap = cap;
  }
  std::atomic<unsigned> bad(0);
  const unsigned W = (unsigned)e->W, H = (unsigned)e->H;
  uint32_t *xy = e->h_xy;
  int64_t *tp = e->h_tp;
  if (aos)
    parallel_ranges(n, [&](int64_t a0, int64_t a1) {
      unsigned acc = 0;
      for (int64_t i = a0; i < a1; i++) {
        const unsigned ex = aos->X(i), ey = aos->Y(i);
        acc |= (unsigned)(ex >= W) | (unsigned)(ey >= H);
        xy[i] = ex | (ey << 16);
        tp[i] = aos->T(i);
      }
      if (acc) bad = 1;
    });
  else
  parallel_ranges(n, [&](int64_t a0, int64_t a1) {
    unsigned acc = 0;
    for (int64_t i = a0; i < a1; i++) {
      acc |= (unsigned)(x[i] >= W) | (unsigned)(y[i] >= H);
      xy[i] = (uint32_t)x[i] | ((uint32_t)y[i] << 16);
      tp[i] = t_ns[i];
    }
    if (acc) bad = 1;
  });
  if (bad.load()) return efail(e, CMX_ERR_EVENT_RANGE, "event coordinates outside the sensor");
  for (cmx_events::Replica &r : e->rep) {
    if (hipSetDevice(r.device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
    if (hipMemcpyAsync(r.d_xy[e->cur] + e->size, xy, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, r.stream) != hipSuccess ||
        hipMemcpyAsync(r.d_t[e->cur] + e->size, tp, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, r.stream) != hipSuccess)
      return efail(e, CMX_ERR_HIP, "upload failed");
  }
  for (cmx_events::Replica &r : e->rep) {
    if (hipSetDevice(r.device) != hipSuccess || hipStreamSynchronize(r.stream) != hipSuccess) return efail(e, CMX_ERR_HIP, "upload failed");
  }
  (void)hipSetDevice(e->device);
  e->h_t.insert(e->h_t.end(), tp, tp + n);  // (the packed timestamps: the AoS form has no t_ns[] of its own)
  e->size += (size_t)n;
  return CMX_OK;
}

// AngVelEstimator::deleteOldEvents (ang_vel_estimator.cpp:149-173): forget everything before a global index
int cmx_events_drop_before(cmx_events *e, int64_t global_index) {
  if (!e) return CMX_ERR_INVALID_ARG;
  if (global_index <= e->first_index) return CMX_OK;
  if (global_index > e->first_index + (int64_t)e->size) return efail(e, CMX_ERR_INVALID_ARG, "index beyond the stored events");
  const size_t k = (size_t)(global_index - e->first_index), keep = e->size - k;
  const int other = 1 - e->cur;
  if (keep) {
    for (cmx_events::Replica &r : e->rep) {
      if (hipSetDevice(r.device) != hipSuccess) return efail(e, CMX_ERR_HIP, "hipSetDevice failed");
      if (hipMemcpyAsync(r.d_xy[other], r.d_xy[e->cur] + k, keep * sizeof(uint32_t), hipMemcpyDeviceToDevice, r.stream) != hipSuccess ||
          hipMemcpyAsync(r.d_t[other], r.d_t[e->cur] + k, keep * sizeof(int64_t), hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
        return efail(e, CMX_ERR_HIP, "compaction failed");
    }
    // the contexts that cut packets / windows from the store use their own non-blocking streams: the compaction is complete
    // on every replica before the buffers are flipped
    //  device-wide: cmx_*_set_*_from leaves an asynchronous copy OUT of the current buffer pending on the consumer's own
    // stream -- after two drops without an evaluation in between the second compaction would write the buffer that copy still reads
    for (cmx_events::Replica &r : e->rep)
      if (hipSetDevice(r.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return efail(e, CMX_ERR_HIP, "compaction failed");
    (void)hipSetDevice(e->device);
  }
  e->h_t.erase(e->h_t.begin(), e->h_t.begin() + (ptrdiff_t)k);
  e->cur = other;
  e->size = keep;
  e->first_index = global_index;
  return CMX_OK;
}

static int store_range(cmx_ctx *c, const cmx_events *e, int64_t first, int64_t count, size_t *off, const cmx_events::Replica **rep) {
  if (!e) return fail(c, CMX_ERR_INVALID_ARG, "null event store");
  if (!c) return CMX_ERR_INVALID_ARG;
  *rep = e->on(c->device);
  if (!*rep || e->W != c->W || e->H != c->H)
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
  const cmx_events::Replica *r = nullptr;
  int rc = store_range(c, e, first, count, &off, &r);
  if (rc) return rc;
  return fe_set_packet_impl(c, count, nullptr, nullptr, e->h_t.data() + off, r->d_xy[e->cur] + off, t_ref_ns, fx, fy, cx, cy,
                            event_batch_size, blur_sigma, contrast_measure);
}
int cmx_backend_set_window_from(cmx_ctx *c, const cmx_events *e, int64_t first, int64_t count, int order, int K,
                                const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed,
                                int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate, double blur_sigma,
                                int contrast_measure, const float *IG) {
  if (is_group(c))
    return group_set_window_from(c, e, first, count, order, K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns,
                                 event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG);
  size_t off = 0;
  const cmx_events::Replica *r = nullptr;
  int rc = store_range(c, e, first, count, &off, &r);
  if (rc) return rc;
  return be_set_window_impl(c, count, nullptr, nullptr, e->h_t.data() + off, r->d_xy[e->cur] + off, r->d_t[e->cur] + off, order,
                            K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size,
                            event_sample_rate, blur_sigma, contrast_measure, IG);
}

// A window cut from a replicated store on a GROUP: member r cuts ITS batch range (the range group_set_window hands it, the
// one-event rule included) from the replica on its own device -- no event crosses the host or a link at hand-over
// (pose_graph_optimizer.cpp:131-165 is the copy this replaces; event_pano_warper.cpp:188-196 the loop being sharded).
int group_member_window_from(cmx_ctx *m, const cmx_events *e, int64_t first, int64_t beg, int64_t end, int order, int K,
                             const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                             int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG) {
  size_t off = 0;
  const cmx_events::Replica *r = nullptr;
  int rc = store_range(m, e, first + beg, end - beg, &off, &r);
  if (rc) return rc;
  return be_set_window_impl(m, end - beg, nullptr, nullptr, e->h_t.data() + off, r->d_xy[e->cur] + off, r->d_t[e->cur] + off, order,
                            K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size, event_sample_rate,
                            blur_sigma, contrast_measure, IG);
}
