// cmx_backend.cpp -- cmx_backend_*: the drop-in for global_contrast_{f,df,fdf}
// (src/backend/global_optim_contrast_gsl_analytical.cpp:17-81 -> event_pano_warper.cpp:167-336 -> global_focus_funcs.cpp:52-80)
// and the once-per-window global-map upkeep (event_pano_warper.cpp:81-126).
#include "cmx_context.hpp"

int cmx_backend_create(cmx_ctx **out, int device, int W, int H, const double *lut, int Wp, int Hp) {
  if (Wp < 4 || Hp < 4 || Wp > 65535 || Hp > 65535 || (long long)Wp * Hp > kMaxPixels) {
    if (out) *out = nullptr;
    return CMX_ERR_INVALID_ARG;
  }
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
  // the clears above ran on the null stream and the context's stream is non-blocking: complete them before anything is queued
  HIP_TRY(c, hipDeviceSynchronize());
  return CMX_OK;
}

int be_set_window_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                              const uint32_t *d_raw, const int64_t *d_t, int order, int K, const double *knots_xyzw,
                              int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                              int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure,
                              const float *IG, const EvAos *aos) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind_device(c);
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
  if (!d_raw && !aos) {
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
      const int64_t t_first = aos ? aos->T(beg) : t_ns[beg], t_last = aos ? aos->T(end - 1) : t_ns[end - 1];
      if (t_last < t_first) { err_kind = CMX_ERR_TIME_ORDER; err_at = beg; return; }
      const long long tb = time_batch_ns(t_first, t_last);
      const long long st = tb - start_ns;
      if (st < 0 || st / dt_ns + order > K) { err_kind = CMX_ERR_SPLINE_RANGE; err_at = tb; return; }
      bt[(size_t)b] = tb;
      if (rate == 1) continue;  // packed below by a flat, vectorisable loop (packed index == event index)
      uint32_t *dst = xy + b * per_batch;
      unsigned acc = 0;
      for (int64_t e = beg; e < end; e += rate) {
        const unsigned ex = aos ? aos->X(e) : (unsigned)x[e], ey = aos ? aos->Y(e) : (unsigned)y[e];
        const int64_t et = aos ? aos->T(e) : t_ns[e];
        acc |= (unsigned)(ex >= sensor_w) | (unsigned)(ey >= sensor_h);
        *dst++ = ex | (ey << 16) | ((et < t_next_win_beg_ns) ? 0x80000000u : 0u);
      }
      if (acc) out_of_range = 1;
    }
  });
  if (rate == 1 && !d_raw && aos)  // straight from the host's records (dvs_msgs::Event): no x[] / y[] / t_ns[] vectors in between
    parallel_ranges(n_packed_total, [&](int64_t a0, int64_t a1) {
      unsigned acc = 0;
      for (int64_t e = a0; e < a1; e++) {
        const unsigned ex = aos->X(e), ey = aos->Y(e);
        acc |= (unsigned)(ex >= sensor_w) | (unsigned)(ey >= sensor_h);
        xy[e] = ex | (ey << 16) | ((uint32_t)(aos->T(e) < t_next_win_beg_ns) << 31);
      }
      if (acc) out_of_range = 1;
    });
  else if (rate == 1 && !d_raw)
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
    rc = check_events(c, n, x, y, t_ns, aos);
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
  // (on the context's stream, not the null stream: a masked stream -- cmx_set_cu_mask -- is a BLOCKING stream, and a null-stream
  //  copy would serialise this hand-over against every other context's null-stream work on the device; `bt` lives until the
  //  synchronisation at the end of this function)
  if (nb && !d_raw) HIP_TRY(c, hipMemcpyAsync(c->d_batch_t, bt.data(), (size_t)nb * sizeof(long long), hipMemcpyHostToDevice, c->stream));
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
    HIP_TRY(c, hipMemcpyAsync(c->d_IG, IG, np * sizeof(float), hipMemcpyHostToDevice, c->stream));
    std::atomic<bool> nz(false);
    parallel_ranges((int64_t)np, [&](int64_t a0, int64_t a1) {
      for (int64_t i = a0; i < a1 && !nz.load(std::memory_order_relaxed); i++)
        if (IG[i] != 0.f) nz = true;
    });
    c->ig_nonzero = nz.load();
  } else {
    HIP_TRY(c, hipMemsetAsync(c->d_IG, 0, np * sizeof(float), c->stream));
    c->ig_nonzero = false;
  }
  HIP_TRY(c, hipMemsetAsync(c->d_alpha, 0, sizeof(double), c->stream));  // on the context's (non-blocking) stream: ordered before its kernels
  c->h_result[kAlphaSlot] = 0.0;  // alpha mirror
  c->first_iter = true;     // setFirstIter(true), pose_graph_optimizer.cpp:293
  comm_reset_xset(c);       // sharded panoramas: the first exchange of a window covers the whole planes
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (nb && d_raw) {
    long long e[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(e, c->d_batch_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (e[0] == CMX_ERR_TIME_ORDER) return fail(c, CMX_ERR_TIME_ORDER, "batch at event %lld spans a negative time interval", e[1]);
    if (e[0] == CMX_ERR_SPLINE_RANGE)
      return fail(c, CMX_ERR_SPLINE_RANGE, "batch time %lld ns outside the support of %d knots (start %lld, dt %lld)", e[1], K,
                  (long long)start_ns, (long long)dt_ns);
  }
  c->n_packed = (int)n_packed_total;
  c->per_batch = per_batch;
  c->nb = nb;
  c->have_data = true;
  c->tb_valid = false;
  c->bin_valid = false;
  return CMX_OK;
}

int cmx_backend_set_window(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                           int order, int K, const double *knots_xyzw, int64_t start_ns, int64_t dt_ns,
                           int num_fixed, int64_t t_next_win_beg_ns, int event_batch_size, int event_sample_rate,
                           double blur_sigma, int contrast_measure, const float *IG) {
  if (is_group(c))
    return group_set_window(c, nullptr, n, x, y, t_ns, order, K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size,
                            event_sample_rate, blur_sigma, contrast_measure, IG);
  return be_set_window_impl(c, n, x, y, t_ns, nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns, num_fixed,
                            t_next_win_beg_ns, event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG);
}

int cmx_backend_set_window_aos(cmx_ctx *c, int64_t n, const void *events, const cmx_aos_layout *layout, int order, int K,
                               const double *knots_xyzw, int64_t start_ns, int64_t dt_ns, int num_fixed, int64_t t_next_win_beg_ns,
                               int event_batch_size, int event_sample_rate, double blur_sigma, int contrast_measure, const float *IG) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  EvAos aos;
  const int rc = make_aos(c, n, events, layout, &aos);
  if (rc) return rc;
  if (is_group(c))
    return group_set_window(c, &aos, n, nullptr, nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns,
                            event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG);
  return be_set_window_impl(c, n, nullptr, nullptr, nullptr, nullptr, nullptr, order, K, knots_xyzw, start_ns, dt_ns, num_fixed,
                            t_next_win_beg_ns, event_batch_size, event_sample_rate, blur_sigma, contrast_measure, IG, &aos);
}

// knot_i <- exp(drot_i) * knot_i for the non-fixed knots (CopyAndIncrementalUpdate, trajectory.cpp:240-263)
static void be_update_knots(cmx_ctx *c, const double *drotv) {
  for (int i = 0; i < c->K; i++) {
    Quat q = c->knots0[i];
    if (i >= c->num_fixed) {
      const double *d = drotv + 3 * (i - c->num_fixed);
      q = q_mul(so3_exp(d[0], d[1], d[2]), q);
    }
    c->h_spline->knots[i] = q;
  }
}

// the time-ordered bearing stream of the gradient gather (once per window)
int be_ensure_time_bearings(cmx_ctx *c) {
  if (c->tb_valid || !c->d_lut2 || c->n_packed <= 0) return CMX_OK;
  int rc = ensure(c, c->d_tb, c->tb_cap, (size_t)2 * c->n_packed);
  if (rc) return rc;
  launch_bearing_stream(c->d_xy, c->d_lut2, c->W, c->n_packed, c->d_tb, c->stream);
  c->tb_valid = true;
  return CMX_OK;
}

// The back end's counterpart of cmx_frontend_prepare: pose table at `drotv_hint` (NULL = zero increments), destination-tile
// sort, chunk table and the bearing streams of a window, queued behind its upload.  A host that owns two contexts prepares
// window k+1 while window k is being solved (pose_graph_optimizer.cpp:244-376 is the loop this sits in).
static int be_prepare_one(cmx_ctx *c, const double *drotv_hint);
int cmx_backend_prepare(cmx_ctx *c, const double *drotv_hint) {
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return be_prepare_one(m, drotv_hint); });
  return be_prepare_one(c, drotv_hint);
}
static int be_prepare_one(cmx_ctx *c, const double *drotv_hint) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_backend_set_window has not succeeded");
  int rc = bind_device(c);
  if (rc) return rc;
  // whatever this call goes on to do, the resident pose table / image no longer stand for "the last evaluation's point" -- cleared
  // BEFORE the early return: a group member whose shard is empty must take the same decision (recompute + exchange) as its peers
  // in the next evaluation, or the collectives no longer match
  c->x_valid = false;
  c->jt_valid = false;
  if (c->splat_mode != 1 || !adjoint_ok(c) || c->n_packed <= 0) return CMX_OK;
  std::vector<double> zero((size_t)3 * (c->K - c->num_fixed > 0 ? c->K - c->num_fixed : 1), 0.0);
  be_update_knots(c, drotv_hint ? drotv_hint : zero.data());
  launch_be_pose_table(*c->h_spline, c->d_batch_t, c->nb, c->order, false, c->d_poseR, c->d_poses, c->stream);
  BeSplatArgs a = be_args(c);
  rc = do_binning(c, nullptr, &a);
  if (rc) return rc;
  if (c->per_batch % 4 == 0) {
    rc = be_ensure_time_bearings(c);
    if (rc) return rc;
  }
  HIP_TRY(c, hipGetLastError());
  c->x_valid = false;  // (the pose table no longer belongs to the last evaluation's point)
  c->jt_valid = false;
  return CMX_OK;
}

static int be_accumulate(cmx_ctx *c, const double *drotv, bool want_grad) {
  yield_to_urgent(c);
  c->timing_tick++;  // see fe_accumulate
  const size_t np = (size_t)c->Wp * c->Hp;
  const int Kopt = c->K - c->num_fixed;
  c->last_adjoint = want_grad && adjoint_ok(c);
  const bool deriv = want_grad && !c->last_adjoint;
  const int P = deriv ? 3 * Kopt : 0;
  int rc = CMX_OK;
  be_update_knots(c, drotv);
  {
    Span sp(c, CMX_T_POSE, /*exact=*/true);
    launch_be_pose_table(*c->h_spline, c->d_batch_t, c->nb, c->order, want_grad || (adjoint_ok(c) && c->reuse_image), c->d_poseR, c->d_poses,
                         c->stream, sp.t0(), sp.t1());
  }
  rc = begin_accum(c, 2 + P, np, P == 0 && adjoint_ok(c) && c->splat_mode == 1);
  if (rc) return rc;
  BeSplatArgs a = be_args(c);
  const bool lds_mode = c->splat_mode == 1 && !deriv;  // what this call would use for any number of events
  const bool use_lds = lds_mode && c->n_packed > 0;
  if (use_lds && (!c->bin_valid || c->last_fallback_frac > kRebinFallbackFrac)) {
    rc = do_binning(c, nullptr, &a);
    if (rc) return rc;
  }
  // tile occupancy: only for the LDS splat into this context's own ping-pong buffers (with a communicator attached the
  // flags are all-reduced with the planes, finish_sharded; planes owned by the caller are exchanged by the caller).
  // Deliberately NOT a function of n_packed: a rank whose shard is empty keeps all-zero flags and takes part in the
  // same collectives as every other rank (cmx_comm.cpp).
  const bool use_flags = lds_mode && c->pingpong_planes > 0 && !c->accum_external;
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
      if (c->deterministic) {
        rc = ensure_fixed(c, 2 * np);
        if (rc) return rc;
        b.fixed = c->d_fixed;
      }
      launch_be_splat_lds(a, b, c->stream, sp.t0(), sp.t1());
    } else {
      launch_be_splat(a, deriv, c->stream, sp.t0(), sp.t1());
    }
  }
  if (use_lds && c->deterministic) launch_fixed_to_float(c->d_fixed, c->d_accum, 2 * np, c->stream);
  c->accum_flagged = use_flags;
  HIP_TRY(c, hipGetLastError());
  c->accum_count = (size_t)(2 + P) * np;
  c->last_P = P;
  c->accumulated = true;
  c->x_valid = true;
  c->jt_valid = false;
  c->gated_pending = false;  // (a gated gradient pass still in flight belongs to the previous point: nobody will ask for it)
  for (int k = 0; k < 3 * Kopt && k < 3 * kMaxKnots; k++) c->last_x[k] = drotv[k];
  return CMX_OK;
}

static int be_accumulate_checked(cmx_ctx *c, const double *drotv, int want_grad);
int cmx_backend_accumulate(cmx_ctx *c, const double *drotv, int want_grad) {
  CMX_NOT_FOR_GROUPS(c, "the split-phase interface");
  return be_accumulate_checked(c, drotv, want_grad);
}
static int be_accumulate_checked(cmx_ctx *c, const double *drotv, int want_grad) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_backend_set_window has not succeeded");
  if (!drotv && c->K > c->num_fixed) return fail(c, CMX_ERR_INVALID_ARG, "null drotv");
  int rc = bind_device(c);
  if (rc) return rc;
  return be_accumulate(c, drotv, want_grad != 0);
}

int be_first_iter(cmx_ctx *c) {
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

static int be_finish_one(cmx_ctx *c, double *contrast, double *grad);
int cmx_backend_finish(cmx_ctx *c, double *contrast, double *grad) {
  CMX_NOT_FOR_GROUPS(c, "the split-phase interface");
  return be_finish_one(c, contrast, grad);
}
static int be_finish_one(cmx_ctx *c, double *contrast, double *grad) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish without accumulate");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  const int P = 3 * (c->K - c->num_fixed);
  if (grad && c->last_P != P && !c->last_adjoint)
    return fail(c, CMX_ERR_STATE, "gradient requested but accumulate ran without it");
  int rc = bind_device(c);
  if (rc) return rc;
  rc = be_first_iter(c);
  if (rc) return rc;
  if (grad && c->last_adjoint) {
    bool served = false;
    c->gate_mode = 0;  // a hint not consumed by a cost-only evaluation does not outlive the next evaluation of any kind
    rc = collect_gated(c, P, contrast, grad, &served);  // the gradient pass may already be in flight (cmx_hint_next_df)
    if (rc || served) return rc;
    rc = run_adjoint(c, P);
  } else if (!grad && speculative_jt_ok(c) && P > 0) {
    rc = finish_cost_only_speculative(c, P);
    if (rc) return rc;
    *contrast = c->h_result[0];
    return CMX_OK;
  } else {
    c->gated_pending = false;
    c->gate_mode = 0;
    rc = run_image_and_finalize(c, grad ? P : 0, nullptr, nullptr);
  }
  if (rc) return rc;
  rc = sync_and_collect(c, true);
  if (rc) return rc;
  *contrast = c->h_result[0];
  if (grad) for (int k = 0; k < P; k++) grad[k] = c->h_result[2 + k];
  return CMX_OK;
}

int cmx_backend_eval(cmx_ctx *c, const double *drotv, double *contrast, double *grad) {
  UrgentScope urgent(c);
  if (is_group(c)) return group_eval(c, drotv, contrast, grad);  // one call, N devices, one contrast / gradient
  return be_eval_one(c, drotv, contrast, grad);
}
int be_eval_one(cmx_ctx *c, const double *drotv, double *contrast, double *grad) {
  const bool sharded = c && c->sharded();
  if (c && c->kind == KIND_BE && drotv && can_reuse(c, drotv, 3 * (c->K - c->num_fixed), grad != nullptr)) {
    c->last_adjoint = true;
    c->reuse_hits++;
    if (sharded) return finish_sharded(c, KIND_BE, false, contrast, grad);
    return be_finish_one(c, contrast, grad);
  }
  int rc = be_accumulate_checked(c, drotv, grad != nullptr);
  if (rc) return rc;
  if (sharded) return finish_sharded(c, KIND_BE, true, contrast, grad);
  return be_finish_one(c, contrast, grad);
}

// ---- global-map upkeep on the device: IG and the visit counts stay resident
static int be_update_map_one(cmx_ctx *c, int max_update_times);
int cmx_backend_update_map(cmx_ctx *c, int max_update_times) {
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return be_update_map_one(m, max_update_times); });  // every member keeps its own replica of the map
  return be_update_map_one(c, max_update_times);
}
static int be_update_map_one(cmx_ctx *c, int max_update_times) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "no evaluation has run in this window (IL_old undefined)");
  int rc = bind_device(c);
  if (rc) return rc;
  launch_update_map(c->d_IG, c->d_accum, c->d_visits, c->Wp * c->Hp, max_update_times, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMX_OK;
}
static int be_mark_visited_one(cmx_ctx *c, const double q[4], int radius);
int cmx_backend_mark_visited(cmx_ctx *c, const double q[4], int radius) {
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return be_mark_visited_one(m, q, radius); });  // every member keeps its own replica of the map
  return be_mark_visited_one(c, q, radius);
}
static int be_mark_visited_one(cmx_ctx *c, const double q[4], int radius) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!q || radius < 0 || radius > 64) return fail(c, CMX_ERR_INVALID_ARG, "bad pose / radius");
  int rc = bind_device(c);
  if (rc) return rc;
  const Mat3 R = q_to_R(Quat{q[0], q[1], q[2], q[3]});
  launch_mark_visited(be_args(c), R.m, c->H, radius, c->d_mask, c->d_visits, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMX_OK;
}
static int be_reset_map_one(cmx_ctx *c);
int cmx_backend_reset_map(cmx_ctx *c) {
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return be_reset_map_one(m); });  // every member keeps its own replica of the map
  return be_reset_map_one(c);
}
static int be_reset_map_one(cmx_ctx *c) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind_device(c);
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
  int rc = bind_device(c);
  if (rc) return rc;
  const size_t np = (size_t)c->Wp * c->Hp;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (IG) HIP_TRY(c, hipMemcpy(IG, c->d_IG, np * sizeof(float), hipMemcpyDeviceToHost));
  if (visits) HIP_TRY(c, hipMemcpy(visits, c->d_visits, np, hipMemcpyDeviceToHost));
  return CMX_OK;
}
static int be_set_map_one(cmx_ctx *c, const float *IG, const unsigned char *visits);
int cmx_backend_set_map(cmx_ctx *c, const float *IG, const unsigned char *visits) {
  if (is_group(c)) return group_all(c, [&](cmx_ctx *m, int) { return be_set_map_one(m, IG, visits); });  // every member keeps its own replica of the map
  return be_set_map_one(c, IG, visits);
}
static int be_set_map_one(cmx_ctx *c, const float *IG, const unsigned char *visits) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  int rc = bind_device(c);
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
  int rc = bind_device(c);
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

// One context's rows, computed into SCRATCH tables from a COPY of the spline description: neither the live pose table (the tile
// sort of a prepared window was built on it) nor h_spline (a prepare may have moved it to its hint) is touched.  The knots are
// those of the last evaluation (last_x), or zero increments before the first one of the window.
static int be_pose_table_rows(cmx_ctx *c, int max_batches, double *R, float *Jcp, int *idx, int64_t *t_batch_ns, int *n_rows) {
  int rc = bind_device(c);
  if (rc) return rc;
  const int n = c->nb < max_batches ? c->nb : max_batches;
  *n_rows = n;
  if (n <= 0) return CMX_OK;
  SplineArgs sp = *c->h_spline;
  for (int i = 0; i < c->K; i++) {
    Quat q = c->knots0[(size_t)i];
    if (c->accumulated && i >= c->num_fixed) {
      const double *d = c->last_x + 3 * (i - c->num_fixed);
      q = q_mul(so3_exp(d[0], d[1], d[2]), q);
    }
    sp.knots[i] = q;
  }
  PoseR *d_r = nullptr;
  PoseEntry *d_e = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_r, (size_t)c->nb * sizeof(PoseR)));
  if (hipMalloc((void **)&d_e, (size_t)c->nb * sizeof(PoseEntry)) != hipSuccess) { (void)hipFree(d_r); return fail(c, CMX_ERR_HIP, "hipMalloc failed"); }
  // the same launch an evaluation issues, Jacobians included
  launch_be_pose_table(sp, c->d_batch_t, c->nb, c->order, true, d_r, d_e, c->stream);
  std::vector<PoseR> hr((size_t)n);
  std::vector<PoseEntry> he((size_t)n);
  std::vector<long long> ht((size_t)n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(hr.data(), d_r, (size_t)n * sizeof(PoseR), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(he.data(), d_e, (size_t)n * sizeof(PoseEntry), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ht.data(), c->d_batch_t, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_r);
  (void)hipFree(d_e);
  if (e != hipSuccess) return fail(c, CMX_ERR_HIP, "pose-table read-back failed: %s", hipGetErrorString(e));
  for (int b = 0; b < n; b++) {
    if (R) memcpy(R + (size_t)9 * b, hr[(size_t)b].R, 9 * sizeof(double));
    if (Jcp) memcpy(Jcp + (size_t)36 * b, he[(size_t)b].Jcp, 36 * sizeof(float));
    if (idx) idx[b] = he[(size_t)b].idx_cp_beg;
    if (t_batch_ns) t_batch_ns[b] = (int64_t)ht[(size_t)b];
  }
  return CMX_OK;
}
int cmx_backend_get_pose_table(cmx_ctx *c, int max_batches, double *R, float *Jcp, int *idx, int64_t *t_batch_ns, int *n_batches) {
  if (!c || c->kind != KIND_BE) return fail(c, CMX_ERR_STATE, "not a back-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_backend_set_window has not succeeded");
  if (max_batches < 0) return fail(c, CMX_ERR_INVALID_ARG, "bad row count");
  if (!is_group(c)) {
    if (n_batches) *n_batches = c->nb;
    int n = 0;
    return be_pose_table_rows(c, max_batches, R, Jcp, idx, t_batch_ns, &n);
  }
  // a group: the members' rows one after the other ARE the window's batches (every member holds whole batches of it, in order);
  // a diagnostic call: the members are read one at a time from the calling thread
  int total = 0, written = 0;
  cmx_ctx *members[64];
  const int nm = group_members(c, members, 64);
  for (int r = 0; r < nm; r++) total += members[r]->nb;
  if (n_batches) *n_batches = total;
  for (int r = 0; r < nm && written < max_batches; r++) {
    int n = 0;
    const int rc = be_pose_table_rows(members[r], max_batches - written, R ? R + (size_t)9 * written : nullptr,
                                      Jcp ? Jcp + (size_t)36 * written : nullptr, idx ? idx + written : nullptr,
                                      t_batch_ns ? t_batch_ns + written : nullptr, &n);
    if (rc) {
      if (r) c->err = members[r]->err;
      (void)bind_device(c);
      return rc;
    }
    written += n;
  }
  return bind_device(c);
}

int cmx_backend_get_alpha(cmx_ctx *c, double *alpha) {
  if (!c || c->kind != KIND_BE || !alpha) return fail(c, CMX_ERR_INVALID_ARG, "bad argument");
  int rc = bind_device(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *alpha = c->h_result[kAlphaSlot];
  return CMX_OK;
}

