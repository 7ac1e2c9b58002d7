// cmx_frontend.cpp -- cmx_frontend_*: the drop-in for local_contrast_{f,df,fdf}
// (src/frontend/local_optim_contrast_gsl.cpp:20-70 -> local_image_warped_events.cpp:10-170 -> local_focus_funcs.cpp:82-120).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cmx_context.hpp"

int cmx_frontend_create(cmx_ctx **out, int device, int W, int H, const double *lut) {
  int rc = create_common(out, KIND_FE, device, W, H, lut);
  if (rc) return rc;
  (*out)->imgW = W;
  (*out)->imgH = H;
  return chain_prealloc(*out);
}

// d_raw != nullptr: the events are already on the device (event store), x / y are unused and t_ns is the store's
// host mirror of the timestamps
int fe_set_packet_impl(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                              const uint32_t *d_raw, int64_t t_ref_ns, double fx, double fy, double cx, double cy,
                              int event_batch_size, double blur_sigma, int contrast_measure, const EvAos *aos) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  int rc = bind_device(c);
  if (rc) return rc;
  c->have_data = false;
  c->accumulated = false;
  c->x_valid = false;
  if (event_batch_size <= 0) return fail(c, CMX_ERR_INVALID_ARG, "event_batch_size must be > 0");
  // computeContrast's switch (local_focus_funcs.cpp:98-109): 1 = mean square, 2 = gradient magnitude, default = variance
  if (contrast_measure != CMX_MEAN_SQUARE && contrast_measure != CMX_GRADIENT_MAGNITUDE) contrast_measure = CMX_VARIANCE;
  if (!d_raw && !aos) {
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
    if (aos)  // straight from the host's records (dvs_msgs::Event): the same words, no x[] / y[] vectors in between
      parallel_ranges(n, [&](int64_t a, int64_t b) {
        unsigned acc = 0;
        for (int64_t i = a; i < b; i++) {
          const unsigned ex = aos->X(i), ey = aos->Y(i);
          acc |= (unsigned)(ex >= W) | (unsigned)(ey >= H);
          xy[i] = ex | (ey << 16);
        }
        if (acc) out_of_range = 1;
      });
    else
    parallel_ranges(n, [&](int64_t a, int64_t b) {
      unsigned acc = 0;
      for (int64_t i = a; i < b; i++) {
        acc |= (unsigned)(x[i] >= W) | (unsigned)(y[i] >= H);
        xy[i] = (uint32_t)x[i] | ((uint32_t)y[i] << 16);
      }
      if (acc) out_of_range = 1;
    });
    if (out_of_range.load()) return check_events(c, n, x, y, t_ns, aos);  // locate and report the offender
  }
  rc = ensure_pinned_dts(c, (size_t)nb);
  if (rc) return rc;
  double *dts = c->h_dts;  // pinned: uploaded asynchronously below (the synchronisation above protects its reuse)
  const double tref = time_to_sec(t_ref_ns);
  std::atomic<int> bad_batch(-1);
  parallel_ranges(nb, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      const int64_t beg = b * event_batch_size;
      const int64_t end = (beg + event_batch_size < n) ? beg + event_batch_size : n;
      const int64_t t_first = aos ? aos->T(beg) : t_ns[beg], t_last = aos ? aos->T(end - 1) : t_ns[end - 1];
      if (t_last < t_first) { bad_batch = (int)b; return; }
      dts[(size_t)b] = time_to_sec(time_batch_ns(t_first, t_last)) - tref;
    }
  }, /*serial_below=*/4096);
  if (bad_batch.load() >= 0) return fail(c, CMX_ERR_TIME_ORDER, "batch %d spans a negative time interval", bad_batch.load());
  rc = ensure(c, c->d_xy, c->xy_cap, (size_t)n);
  if (rc) return rc;
  rc = ensure(c, c->d_batch_dt, c->batch_cap, (size_t)nb);
  if (rc) return rc;
  if (n) {
    if (d_raw) HIP_TRY(c, hipMemcpyAsync(c->d_xy, d_raw, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(c, hipMemcpyAsync(c->d_xy, xy, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    // both uploads are stream-ordered in front of everything the evaluations launch; nobody waits for them here (the 4 MB of
    // a 1M-event packet cross PCIe while the caller is already issuing the first evaluation)
    HIP_TRY(c, hipMemcpyAsync(c->d_batch_dt, dts, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  c->n_packed = (int)n;
  c->per_batch = event_batch_size;
  c->nb = nb;
  c->have_data = true;
  c->tb_valid = false;
  c->bin_valid = false;
  return CMX_OK;
}

int cmx_frontend_set_packet(cmx_ctx *c, int64_t n, const uint16_t *x, const uint16_t *y, const int64_t *t_ns,
                            int64_t t_ref_ns, double fx, double fy, double cx, double cy, int event_batch_size,
                            double blur_sigma, int contrast_measure) {
  return fe_set_packet_impl(c, n, x, y, t_ns, nullptr, t_ref_ns, fx, fy, cx, cy, event_batch_size, blur_sigma, contrast_measure);
}

int cmx_frontend_set_packet_aos(cmx_ctx *c, int64_t n, const void *events, const cmx_aos_layout *layout, int64_t t_ref_ns, double fx,
                                double fy, double cx, double cy, int event_batch_size, double blur_sigma, int contrast_measure) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  EvAos aos;
  const int rc = make_aos(c, n, events, layout, &aos);
  if (rc) return rc;
  return fe_set_packet_impl(c, n, nullptr, nullptr, nullptr, nullptr, t_ref_ns, fx, fy, cx, cy, event_batch_size, blur_sigma,
                            contrast_measure, &aos);
}

// CMX_FUSE_TRACE (diagnostics): looked up during the first few hundred evaluations of the process only (tools/fuse_trace.py sets it after its warm-up)
static const char *fuse_trace_path() {
  static const char *path = nullptr;
  static int looks = 0;
  if (!path && looks < 1024) { looks++; path = getenv("CMX_FUSE_TRACE"); }
  return path;
}

// The adjoint image pass can ride inside the splat launch (FusedArgs, cmx_internal.hpp): gradient evaluations of the production
// path at the reference's blur (sigma 1 -> radius 4, register-resident operator rows), context-owned ping-pong planes, nothing
// between splat and blur (no communicator), no bitwise-reproducibility promise to keep.
static bool fe_fuse_ok(const cmx_ctx *c, int nplanes, bool use_lds, bool allow_fuse) {
  return allow_fuse && c->fused_image && use_lds && nplanes == 1 && c->last_adjoint && adjoint_ok(c) && c->composite_image &&
         c->radius == 4 && c->d_Mx && c->d_My && c->Mx_radius == 4 && !c->deterministic && !c->sharded() && !c->accum_external &&
         (!c->chain_active || c->fuse_macc) && c->pingpong_planes > 0 && c->fused_bin_id != 0 && c->fused_bin_id == c->binning_id;
}

// ... and the gather + finalize as well (FUSE = 2): a plain gradient evaluation of the production path whose finalize would have been
// the gather's tail anyway
static bool fe_full_ok(const cmx_ctx *c) {
  return c->fused_full && c->streams_valid && c->d_cx && c->d_cy && c->d_gacc && c->d_tail_counters && c->tail_finalize == 1 &&
         c->ticket_wait && c->measure != CMX_GRADIENT_MAGNITUDE && !c->chain_active && c->d_ftile_done && c->n_packed > 0;
}

// ... or as ONE launch of the chunk workgroups alone (FUSE = 3, cmx_selfserve.hpp): the same evaluations, once the chunk table's exact
// length is known and all of its workgroups are resident at once on this device
static bool fe_self_ok(const cmx_ctx *c) {
  return c->fused_self && c->streams_valid && c->d_cx && c->d_cy && c->d_gacc && c->d_tail_counters && c->tail_finalize == 1 &&
         c->ticket_wait && c->measure != CMX_GRADIENT_MAGNITUDE && !c->chain_active && c->d_ftile_done && c->n_packed > 0 &&
         c->nchunks_exact && c->nchunks > 0 && c->nchunks <= fe_selfserve_capacity();
}

int fe_accumulate(cmx_ctx *c, const double omega[3], int nplanes, bool allow_fuse, bool allow_full) {
  yield_to_urgent(c);
  c->timing_tick++;  // every span of this evaluation (accumulate and finish) samples, or none does
  const size_t np = (size_t)c->W * c->H;
  // what the buffer the previous evaluation voted into looks like (it becomes this evaluation's ping-pong partner)
  const unsigned prev_votes_bin = c->votes_bin_id;
  const bool prev_in_reach = c->last_used_lds && c->last_fallback_flags == 0u && !c->fallback_pending;  // (no vote beyond kFuseReach)
  const float *prev_accum = c->d_accum;
  int rc = begin_accum(c, nplanes, np, nplanes == 1 && adjoint_ok(c) && c->splat_mode == 1);
  if (rc) return rc;
  FeSplatArgs a = fe_args(c, omega);
  for (int k = 0; k < 3; k++) c->last_x[k] = omega[k];
  const bool use_lds = c->splat_mode == 1 && nplanes == 1 && c->n_packed > 0;
  if (use_lds && (!c->bin_valid || c->force_rebin || c->last_fallback_frac > kRebinFallbackFrac)) {
    rc = do_binning(c, &a, nullptr);
    if (rc) return rc;
  }
  c->fused_done = false;
  c->fused_full_done = false;
  c->votes_bin_id = use_lds ? c->binning_id : 0u;
  {
    Span sp(c, CMX_T_SPLAT, /*exact=*/true);
    c->last_used_lds = use_lds;
    if (use_lds) c->fallback_pending = true;
    if (use_lds) {
      BinnedEvents b = binned(c);
      if (c->deterministic) {
        rc = ensure_fixed(c, np);
        if (rc) return rc;
        b.fixed = c->d_fixed;
      }
      FusedArgs f{};
      const bool fuse = fe_fuse_ok(c, nplanes, use_lds, allow_fuse);
      if (fuse) {
        float *jt_before = c->d_itilde;
        rc = ensure(c, c->d_itilde, c->itilde_cap, np);
        if (rc) return rc;
        if (c->d_itilde != jt_before) HIP_TRY(c, hipMemsetAsync(c->d_itilde, 0, c->itilde_cap * sizeof(float), c->stream));
        f.tiles_x = c->fused_tiles_x;
        f.tiles_y = c->fused_tiles_y;
        f.nbr_expected = c->d_fnbr_expected;
        f.nbr_cnt = c->d_fnbr_cnt;
        memcpy(f.taps, c->taps, sizeof(f.taps));
        f.Mx = c->d_Mx;
        f.My = c->d_My;
        f.jt = c->d_itilde;
        f.partials = c->d_fpartials;
        f.macc = c->chain_active ? c->fuse_macc : nullptr;
        if (c->d_accum_alt && !c->alt_clean) {
          // ping-pong: the tiles' passes clear the partner.  They cover every pixel the partner's votes can have reached only if
          // those votes were made under THIS chunk table and stayed within reach of their tiles; otherwise one memset clears it.
          // (slots of a device-driven solve are queued AHEAD of their predecessors' results: they assume reach -- a slot that reports
          //  otherwise ends the chain, cmx_chain.cpp, and both buffers are cleared before the next evaluation)
          const bool covered = c->d_accum_alt == prev_accum && prev_votes_bin == c->binning_id &&
                               (prev_in_reach || (c->chain_active && c->last_used_lds && c->last_fallback_flags == 0u));
          if (covered) f.zero_ptr = c->d_accum_alt;
          else HIP_TRY(c, hipMemsetAsync(c->d_accum_alt, 0, (size_t)c->pingpong_planes * np * sizeof(float), c->stream));
          c->alt_clean = true;  // stream-ordered: clean by the time the next accumulate's splat runs
          c->alt_flagged = false;
        }
        const bool self = allow_full && fe_self_ok(c);
        if (self || (allow_full && fe_full_ok(c))) {  // ONE launch: gather workgroups and the finalize step behind the strips
          const int NT = 512, n = c->n_packed;
          int per = ((n + 399) / 400 + NT - 1) / NT * NT;
          per = per < NT ? NT : (per > 4 * NT ? 4 * NT : per);
          if (self) {  // ... or the chunk workgroups themselves run the passes and gather their own events
            f.self_serve = 1;
            c->fused_self_evals++;
          } else {
            f.gather_per_block = per;
            f.gather_blocks = (n + per - 1) / per;
          }
          if (++c->fuse_seq == 0u) c->fuse_seq = 1u;
          f.seq = c->fuse_seq;
          f.tile_done = c->d_ftile_done;
          f.tiles_done = c->d_ftiles_done;
          f.n_active = c->d_fn_active;
          f.cx = c->d_cx;
          f.cy = c->d_cy;
          f.gacc = c->d_gacc;
          f.gacc_stride = kGaccStride;
          f.tail_counters = c->d_tail_counters;
          f.result = result_ptr(c);
          f.ticket = ++c->ticket_issued;
          c->ticket_nout = 5;
          f.npix = (double)np;
          f.measure = c->measure;
          c->fused_full_done = true;
          c->fused_full_evals++;
        }
        // (diagnostics, read once per process -- a getenv per evaluation is ~0.1 us of the host's turn-around; tools/fuse_trace.py sets
        //  its variables before the library is loaded or re-reads through CMX_FUSE_TRACE's presence at first use)
        static const char *const env_dbg = getenv("CMX_FUSE_DEBUG");
        if (env_dbg) f.debug = atoi(env_dbg);
        if (fuse_trace_path()) {  // diagnostics: per-workgroup wall-clock stamps of this launch (tools/fuse_trace.py)
          const size_t nwg = (size_t)b.nchunks + (size_t)f.tiles_x * f.tiles_y * kFuseStrips + (size_t)f.gather_blocks;
          rc = ensure(c, c->d_fuse_trace, c->fuse_trace_cap, 8 * nwg);
          if (rc) return rc;
          HIP_TRY(c, hipMemsetAsync(c->d_fuse_trace, 0, 8 * nwg * sizeof(unsigned long long), c->stream));
          f.trace = c->d_fuse_trace;
          c->fuse_trace_n = nwg;
        }
        c->fused_done = true;
        c->fused_evals++;
      }
      launch_fe_splat_lds(a, b, c->stream, sp.t0(), sp.t1(), fuse ? &f : nullptr);
    } else {
      launch_fe_splat(a, nplanes > 1, c->stream, sp.t0(), sp.t1());
    }
  }
  if (use_lds && c->deterministic) launch_fixed_to_float(c->d_fixed, c->d_accum, np, c->stream);
  HIP_TRY(c, hipGetLastError());
  c->accum_count = nplanes * np;
  c->last_P = nplanes - 1;
  c->accumulated = true;
  c->x_valid = true;
  c->jt_valid = c->fused_done;  // (the fused pass leaves Jt and its moment rows exactly as a speculative image pass would)
  if (c->fused_done) {
    c->adj_fused = true;
    c->adj_direct = true;
    c->adj_tile_count = nullptr;
  }
  c->gated_pending = false;  // (a gated gradient pass still in flight belongs to the previous point: nobody will ask for it)
  return CMX_OK;
}

// Everything a packet's FIRST evaluation would do before its first vote -- the destination-tile sort at `omega_hint`, the
// tile-ordered bearing / dt streams, the chunk table -- queued now, behind the packet's upload, without waiting for any of
// it.  With two contexts a host prepares packet k+1 on one while packet k is being solved on the other: the GPU runs the
// upload and the sort in the shadow of the solve (one context fills < 40 % of the chip), and the solve of packet k+1 starts
// with its first evaluation at full speed.  The hint only decides which votes find their LDS window (speed); results do
// not depend on it.  Reference counterpart: the work between getEventSubset and the first cost evaluation of
// AngVelEstimator::handleEvents' optimisation (src/frontend/ang_vel_estimator.cpp:68-147), which the CPU path does not have.
int cmx_frontend_prepare(cmx_ctx *c, const double omega_hint[3]) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_frontend_set_packet has not succeeded");
  if (!omega_hint) return fail(c, CMX_ERR_INVALID_ARG, "null omega");
  int rc = bind_device(c);
  if (rc) return rc;
  if (c->splat_mode != 1 || !adjoint_ok(c) || c->n_packed <= 0) return CMX_OK;  // nothing this configuration sorts
  FeSplatArgs a = fe_args(c, omega_hint);
  rc = do_binning(c, &a, nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}

// allow_fuse: the caller runs the gather of this very point next and nothing else touches the planes in between (cmx_frontend_eval);
// the split-phase entry point never fuses -- its caller may exchange the planes with other ranks before the blur
static int fe_accumulate_checked(cmx_ctx *c, const double omega[3], int want_grad, bool allow_fuse, bool allow_full = false) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_frontend_set_packet has not succeeded");
  if (!omega) return fail(c, CMX_ERR_INVALID_ARG, "null omega");
  int rc = bind_device(c);
  if (rc) return rc;
  c->last_adjoint = want_grad && adjoint_ok(c);
  return fe_accumulate(c, omega, (want_grad && !c->last_adjoint) ? 4 : 1, allow_fuse, allow_full);
}
int cmx_frontend_accumulate(cmx_ctx *c, const double omega[3], int want_grad) { return fe_accumulate_checked(c, omega, want_grad, false); }

int cmx_frontend_finish(cmx_ctx *c, double *contrast, double *grad) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish without accumulate");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  if (grad && c->last_P != 3 && !c->last_adjoint)
    return fail(c, CMX_ERR_STATE, "gradient requested but accumulate ran without it");
  int rc = bind_device(c);
  if (rc) return rc;
  if (grad && c->last_adjoint) {
    bool served = false;
    c->gate_mode = 0;  // a hint not consumed by a cost-only evaluation does not outlive the next evaluation of any kind
    rc = collect_gated(c, 3, contrast, grad, &served);  // the gradient pass may already be in flight (cmx_hint_next_df)
    if (rc || served) return rc;
    const bool fused = c->fused_done;
    if (!c->fused_full_done) rc = run_adjoint(c, 3);  // (one-launch evaluation: gather and finalize are already in flight)
    if (!rc && fused) {
      // the fused image pass is exact only while every vote lands within reach of its chunk's tile (the tiles' arrival counts cover
      // kFuseReach = 56 px around it; votes beyond that, or a tile that gave up waiting, raise a flag in the fallback word): such
      // an evaluation is repeated after a fresh sort at these very parameters -- every vote is then inside its own tile -- and,
      // should that not do, through the separate launches, which are exact for any parameters
      for (int attempt = 0;; attempt++) {
        rc = sync_and_collect(c, true);
        if (rc) return rc;
        if (const char *path = fuse_trace_path()) {
          if (c->d_fuse_trace && c->fuse_trace_n) {
            std::vector<unsigned long long> h(8 * c->fuse_trace_n);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(h.data(), c->d_fuse_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fp); fclose(fp); }
          }
        }
        const unsigned flags = fallback_flags(c->h_result[kFallbackSlot]);
        if (!flags || !c->fused_done) break;
        c->fused_redos++;
        c->force_rebin = true;
        if (flags & kFuseIncomplete) {  // a tile gave up waiting: late arrivals are still on its counter
          c->fused_timeouts++;
          // ... and passes that never ran have not cleared their tiles of the ping-pong partner: neither buffer is known to be clean
          c->alt_clean = false;
          c->accum_clean = false;
          // the self-service form needs every workgroup resident at once: a context that shares the device with other work and keeps
          // running into the bounded waits goes back to the two-launch form for good
          if (c->fused_full_done && c->fused_self && ++c->fused_self_strikes >= 3) c->fused_self = false;
          HIP_TRY(c, hipMemsetAsync(c->d_fnbr_cnt, 0, c->fcnt_cap * sizeof(unsigned), c->stream));
        }
        double om[3] = {c->last_x[0], c->last_x[1], c->last_x[2]};
        const bool again_fused = attempt == 0 && !(flags & kFuseIncomplete);
        rc = fe_accumulate(c, om, 1, /*allow_fuse=*/again_fused, /*allow_full=*/again_fused);
        if (rc) return rc;
        if (!c->fused_full_done) {
          rc = run_adjoint(c, 3);
          if (rc) return rc;
        }
      }
      *contrast = c->h_result[0];
      for (int k = 0; k < 3; k++) grad[k] = c->h_result[2 + k];
      return CMX_OK;
    }
  } else if (!grad && speculative_jt_ok(c)) {
    rc = finish_cost_only_speculative(c, 3);
    if (rc) return rc;
    *contrast = c->h_result[0];
    return CMX_OK;
  } else {
    c->gated_pending = false;
    c->gate_mode = 0;
    rc = run_image_and_finalize(c, grad ? 3 : 0, nullptr, nullptr);
  }
  if (rc) return rc;
  rc = sync_and_collect(c, true);
  if (rc) return rc;
  *contrast = c->h_result[0];
  if (grad) for (int k = 0; k < 3; k++) grad[k] = c->h_result[2 + k];
  return CMX_OK;
}

int cmx_frontend_eval(cmx_ctx *c, const double omega[3], double *contrast, double *grad) {
  UrgentScope urgent(c);
  const bool sharded = c && c->sharded();
  if (c && c->kind == KIND_FE && omega && can_reuse(c, omega, 3, grad != nullptr)) {
    c->last_adjoint = true;  // image of this very point is resident: adjoint blur + gather only
    c->reuse_hits++;
    if (sharded) return finish_sharded(c, KIND_FE, false, contrast, grad);
    return cmx_frontend_finish(c, contrast, grad);
  }
  int rc = fe_accumulate_checked(c, omega, grad != nullptr, /*allow_fuse=*/!sharded, /*allow_full=*/!sharded && grad != nullptr);
  if (rc) return rc;
  if (sharded) return finish_sharded(c, KIND_FE, true, contrast, grad);
  return cmx_frontend_finish(c, contrast, grad);
}

int cmx_frontend_get_iwe(cmx_ctx *c, const double omega[3], int blur, float *iwe, float *deriv) {
  if (!c || c->kind != KIND_FE) return fail(c, CMX_ERR_STATE, "not a front-end context");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "cmx_frontend_set_packet has not succeeded");
  if (!omega || !iwe) return fail(c, CMX_ERR_INVALID_ARG, "null argument");
  int rc = bind_device(c);
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

