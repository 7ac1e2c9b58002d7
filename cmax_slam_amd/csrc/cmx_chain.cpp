// cmx_chain.cpp -- device-driven front-end solve: the FR-CG line search runs ahead of the host.
//
// A host-driven solve pays, after EVERY evaluation, the way to the host and back: finalize -> completion ticket over PCIe ->
// the optimiser's arithmetic -> next launch (~5.7 us of a 25-40 us evaluation).  Everything the optimiser
// does between two evaluations is a closed form of numbers the finalize step has just reduced (GSL's take_step /
// intermediate_point / minimize, the Fletcher-Reeves direction update, the reference's stopping rules,
// src/frontend/local_optim_contrast_gsl.cpp:134-215).  So the finalize step runs it: the FR-CG state machine
// (cmx_frcg_sm.hpp) lives in device memory, the last-arriving workgroup of an evaluation's last kernel feeds it the cost /
// gradient, and it writes the next evaluation point where the next evaluation's kernels -- queued by the host one slot
// ahead -- read it.  One SLOT = one evaluation point:
//     splat (omega from device memory)  ->  image pass + cost finalize + machine step  ->  gradient pass, gated by the machine
//     (+ its finalize + machine step)
// The host keeps kAhead slots queued, receives every finalize's result block in mapped memory, REPLAYS the same machine on
// the reported cost / gradient and compares its next point with the device's, bit for bit.  On agreement the device is simply
// ahead; on any disagreement (the one known source: pow(r, 2.0) of glibc vs r*r on the device, ~0.08 % of direction updates)
// the host stops the chain and continues the solve itself from its own state -- results are those of the host-driven solve
// either way.  Slots queued beyond the end of the solve return on their first instruction (a flag in device memory).
#include "cmx_context.hpp"

namespace {

constexpr int kAhead = 2;        // slots kept queued ahead of the one whose results the host is waiting for
constexpr int kRingSlots = 8;    // ring of result blocks (2 per slot); > kAhead + 1
constexpr int kBlock = 4096;     // doubles per result block (the layout of the one-evaluation result buffer)

struct SlotTickets { unsigned long long a = 0, g = 0; int nout_a = 0, nout_g = 0; bool gated = false, self_gating = false, fused = false; };

int ensure_chain_buffers(cmx_ctx *c) {
  if (c->d_chain) return CMX_OK;
  HIP_TRY(c, hipMalloc((void **)&c->d_chain, sizeof(ChainDev)));
  HIP_TRY(c, hipHostMalloc((void **)&c->h_chain_ring, (size_t)2 * kRingSlots * kBlock * sizeof(double), hipHostMallocMapped));
  HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_chain_ring, c->h_chain_ring, 0));
  memset(c->h_chain_ring, 0, (size_t)2 * kRingSlots * kBlock * sizeof(double));
  HIP_TRY(c, hipHostMalloc((void **)&c->h_chain_init, 2 * sizeof(ChainDev), hipHostMallocMapped));
  HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_chain_init, c->h_chain_init, 0));
  return CMX_OK;
}

// what the chain needs from the configuration: the production path of a front-end context, whole evaluation on this GPU
bool chain_eligible(const cmx_ctx *c) {
  return c->kind == KIND_FE && c->chain_solve && c->have_data && c->n_packed > 0 && c->splat_mode == 1 && adjoint_ok(c) &&
         !c->deterministic && !c->sharded() && !c->accum_external && !c->gsum_external && c->ticket_wait && c->tail_finalize == 1 &&
         c->d_tail_counters && c->d_gacc && c->reuse_image && c->gated_df && !c->timing &&
         c->measure != CMX_GRADIENT_MAGNITUDE;
}

int queue_slot(cmx_ctx *c, int slot, SlotTickets *t) {
  const int r = slot % kRingSlots;
  c->chain_block_a = c->d_chain_ring + (size_t)(2 * r) * kBlock;
  c->chain_block_g = c->d_chain_ring + (size_t)(2 * r + 1) * kBlock;
  const double zero[3] = {0, 0, 0};  // (the kernels read omega from device memory; last_x is not meaningful inside a chain)
  c->last_adjoint = true;
  int rc = fe_accumulate(c, zero, 1);
  if (rc) return rc;
  c->gate_arm = true;
  rc = run_adjoint(c, 3, /*phase=*/3);  // image pass + cost finalize + machine step (decides the gate)
  c->gate_arm = false;
  if (rc) return rc;
  t->a = c->ticket_issued;
  t->nout_a = c->ticket_nout;
  c->gated_pending = false;
  rc = run_adjoint(c, 3, /*phase=*/4);  // gradient pass behind the gate (+ finalize + machine step)
  if (rc) return rc;
  t->gated = c->gated_pending;
  c->gated_pending = false;
  t->g = c->ticket2_issued;
  t->nout_g = c->ticket2_nout;
  c->chain_slots++;
  return CMX_OK;
}

// Self-gating slots (the production shape: blur radius 4, composite image pass, per-event streams): the image pass runs NO
// finalize -- its tiles' moments go to accumulator rows -- and the launch behind it decides by itself whether it is the gradient
// pass (fe_gather_kernel<2>).  One result block per slot: contrast, [gradient], the machine's decisions and next point.
bool self_gating_ok(const cmx_ctx *c) {
  return c->chain_self_gating && c->radius == 4 && c->composite_image && c->d_Mx && c->d_My && c->Mx_radius == 4 && c->d_lut2 &&
         c->d_cx && c->d_cy;
}
int queue_slot_self_gating(cmx_ctx *c, int slot, SlotTickets *t) {
  const int W = c->imgW, H = c->imgH;
  const size_t np = (size_t)W * H;
  const int r = slot % kRingSlots;
  const double zero[3] = {0, 0, 0};
  const bool first = c->chain_first;              // warm start: omega as kernel arguments, no end-of-solve flag, machine from the host
  const double *x = first ? c->chain_x0 : zero;
  const unsigned par = (c->chain_seq + (unsigned)slot) & 1u;  // moment rows this slot adds to (cleared by the slot before it)
  c->last_adjoint = true;
  // splat (omega from device memory; re-sorts first when due).  Round 6: the image pass rides inside this launch when the fused
  // form applies (FusedArgs) -- its tiles' moments go to the same accumulator rows, and the slot is two launches instead of three
  c->fuse_macc = &c->d_chain->macc[par][0][0];
  int rc = fe_accumulate(c, x, 1, /*allow_fuse=*/true);
  c->fuse_macc = nullptr;
  if (rc) return rc;
  if (!c->streams_valid || !c->bin_valid) return fail(c, CMX_ERR_STATE, "self-gating slot without the tile-ordered streams");
  float *jt_before = c->d_itilde;
  rc = ensure(c, c->d_itilde, c->itilde_cap, np);
  if (rc) return rc;
  if (c->d_itilde != jt_before) HIP_TRY(c, hipMemsetAsync(c->d_itilde, 0, c->itilde_cap * sizeof(float), c->stream));
  // ---- image pass: B = G*I moments -> accumulator rows of buffer (slot & 1), Jt = G^T G I, clears the ping-pong partner
  ImgAdjArgs ia{};
  ImgArgs &a = ia.img;
  ia.Mx = c->d_Mx; ia.My = c->d_My;
  ia.jt = c->d_itilde;
  a.W = W; a.H = H; a.r = c->radius;
  memcpy(a.taps, c->taps, sizeof(a.taps));
  a.src_a = c->d_accum;
  a.P = 0;
  a.tiles_x = image_adjoint_tiles_x(W);
  a.nblk = image_adjoint_tiles(W, H);
  a.tiles_y = (H + kTileY - 1) / kTileY;
  if (!c->fused_done && c->pingpong_planes > 0 && c->d_accum_alt && !c->alt_clean) {
    a.zero_ptr = c->d_accum_alt;
    a.zero_planes = c->pingpong_planes;
    c->alt_clean = true;
  }
  a.skip = first ? nullptr : &c->d_chain->done;
  a.macc = &c->d_chain->macc[par][0][0];
  if (!c->fused_done) launch_image_adjoint(ia, c->stream);
  // ---- the self-gating launch: cost finalize + machine step, and the gradient pass when the machine's test says so
  FeGatherArgs g{};
  g.ev = fe_args(c, x);
  g.itilde = c->d_itilde;
  g.gpartials = nullptr;
  g.cx = c->d_cx; g.cy = c->d_cy; g.r = c->radius;
  g.sxy = c->d_sxy; g.sbatch = c->d_sbatch; g.sb = c->d_sb; g.sdt = c->d_sdt;
  FinalizeArgs &f = g.tail.fin;
  f.P = 0;
  f.nblk = a.nblk;
  f.measure = c->measure;
  f.npix = (double)np;
  f.result = c->d_chain_ring + (size_t)(2 * r) * kBlock;
  f.gP = 3;
  f.mu_free = 1;
  f.direct = 1;
  f.gacc = c->d_gacc;
  f.gacc_stride = kGaccStride;
  f.fallback = c->d_fallback;
  f.macc = a.macc;
  f.macc_clear = &c->d_chain->macc[par ^ 1u][0][0];
  f.nout_pad = 2 + 3;
  f.chain.sm = &c->d_chain->sm;
  f.chain.x_req = c->d_chain->x_req;
  f.chain.done = &c->d_chain->done;
  f.chain.abort_flag = &c->d_chain->abort_flag;
  f.chain.stage = 2;
  f.chain.gate_cur = &c->d_chain->gate[par];
  f.chain.gate_next = &c->d_chain->gate[par ^ 1u];
  f.chain.sm_src = first ? &c->d_chain_init[c->chain_init_sel].sm : nullptr;
  f.ticket = ++c->ticket_issued;
  g.tail.counters = c->d_tail_counters;
  launch_fe_gather(g, c->stream);
  HIP_TRY(c, hipGetLastError());
  t->a = f.ticket;
  t->nout_a = f.nout_pad + kChainExtra;
  t->self_gating = true;
  t->fused = c->fused_done;
  t->gated = true;
  c->jt_valid = false;
  c->chain_slots++;
  return CMX_OK;
}

// wait for a result block of the ring; falls back to the stream when the spin budget runs out
int wait_block(cmx_ctx *c, const double *h_block, unsigned long long ticket, int nout) {
  if (spin_for_ticket(h_block, ticket, nout)) return CMX_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (spin_for_ticket(h_block, ticket, nout)) return CMX_OK;
  return fail(c, CMX_ERR_HIP, "a chained evaluation ended without its finalize step (ticket %llu)", ticket);
}

bool same_bits(const double *a, const double *b, int n) { return memcmp(a, b, sizeof(double) * (size_t)n) == 0; }

}  // namespace

int chain_prealloc(cmx_ctx *c) { return ensure_chain_buffers(c); }

int chain_run_frontend(cmx_ctx *c, FrcgSM &hs, bool *completed) {
  *completed = false;
  if (!c || !chain_eligible(c) || hs.n != 3) return CMX_OK;
  int rc = bind_device(c);
  if (rc) return rc;
  rc = ensure_chain_buffers(c);
  if (rc) return rc;
  // a gated pass of an earlier evaluation may still be queued: nothing of it is wanted
  c->gated_pending = false;
  c->gate_mode = 0;
  // ---- the machine's initial state -> device: ONE stream-ordered copy in front of the first slot.  Two pinned staging blocks
  // alternate, so the copy of the previous solve (long finished: its results were waited for) is never overwritten in flight
  // by the one after next.
  // Warm start (the previous device-driven solve ended normally on self-gating slots): no copy at all -- the first slot's kernels
  // take omega as arguments and its finalizing workgroup reads the machine's ~60 words from the pinned block (one PCIe read,
  // ~1.5 us inside the slot's tail, against a 4 us copy kernel and a launch boundary in front of the solve).
  const bool warm = c->chain_warm && self_gating_ok(c) && hs.gate_mode == 4 && hs.req_at_x;
  c->chain_warm = false;  // true again only if THIS solve ends normally
  {
    ChainDev *st = c->h_chain_init + (c->chain_init_sel ^= 1);
    memset(st, 0, sizeof(ChainDev));
    sm_to_fixed<kChainMaxN>(hs, st->sm);
    for (int k = 0; k < 3; k++) st->x_req[k] = c->chain_x0[k] = hs.x[k];
    st->done = 0;
    for (int q = 0; q < 2; q++) { st->gate[q].thr = hs.gate_thr; st->gate[q].mode = hs.gate_mode; }  // (a cold start's first slot has parity 0)
    if (!warm) {
      HIP_TRY(c, hipMemcpyAsync(c->d_chain, st, sizeof(ChainDev), hipMemcpyHostToDevice, c->stream));
      c->chain_seq = 0;
    } else {
      c->chain_warm_starts++;
    }
  }
  bool all_self_gating = true;
  SlotTickets tick[kRingSlots];
  int queued = 0, consumed = 0;
  bool diverged = false, unsupported = false, fuse_incomplete = false;
  c->chain_active = true;
  c->chain_solves++;
  double g[3];
  while (!sm_done(hs)) {
    while (queued < consumed + kAhead) {
      tick[queued % kRingSlots] = SlotTickets{};
      c->chain_first = warm && queued == 0;
      rc = self_gating_ok(c) ? queue_slot_self_gating(c, queued, &tick[queued % kRingSlots]) : queue_slot(c, queued, &tick[queued % kRingSlots]);
      c->chain_first = false;
      if (rc) break;
      all_self_gating = all_self_gating && tick[queued % kRingSlots].self_gating;
      if (!tick[queued % kRingSlots].gated) { unsupported = true; }  // no gated gradient pass in this configuration
      queued++;
      if (unsupported) break;
    }
    if (rc || unsupported) break;
    const SlotTickets &t = tick[consumed % kRingSlots];
    const double *ba = c->h_chain_ring + (size_t)(2 * (consumed % kRingSlots)) * kBlock;
    const double *bg = ba + kBlock;
    rc = wait_block(c, ba, t.a, t.nout_a);
    if (rc) break;
    if (!c->nchunks_exact && c->bin_valid && c->h_nchunks) {  // once per binning: launch exactly the chunks that exist
      const unsigned long long w = *reinterpret_cast<volatile unsigned long long *>(c->h_nchunks);
      if ((unsigned)(w >> 32) == c->binning_id) {
        const int nch = (int)(unsigned)(w & 0xffffffffull);
        if (nch >= 0 && nch <= c->nchunks) c->nchunks = nch;
        c->nchunks_exact = true;
      }
    }
    if (c->n_packed > 0) c->last_fallback_frac = fallback_count(ba[kFallbackSlot]) / (double)c->n_packed;  // drives the re-sort of the next slot queued
    c->fallback_pending = false;
    if (const unsigned flags = t.fused ? fallback_flags(ba[kFallbackSlot]) : 0u) {
      // a fused slot whose votes went beyond the reach its tiles' arrival counts cover (or a tile that gave up waiting): its numbers --
      // and what the device's machine made of them -- are not this point's.  Nothing of the block is fed to the host's machine: it
      // takes over from its own state, re-evaluates the point through the ordinary path (which sorts again first).
      c->last_fallback_flags = flags;
      c->force_rebin = true;
      c->fused_redos++;
      fuse_incomplete = (flags & kFuseIncomplete) != 0;
      if (fuse_incomplete) c->fused_timeouts++;
      diverged = true;
      break;
    }
    c->last_fallback_flags = 0u;
    if (t.self_gating) {  // one block: contrast, gradient (when the launch computed it), decisions, next point
      const int ext = t.nout_a - kChainExtra;
      if (((int)ba[ext + 2] & 2) != 0) { diverged = true; break; }  // the launch's workgroups and the machine disagreed: nothing fed
      const bool dev_need = ba[ext] != 0.0;
      const bool need = sm_cost(hs, -ba[0]);
      if (need != dev_need) { diverged = true; break; }
      if (c->chain_test == 2 && need && consumed == 2) { diverged = true; break; }  // (test hook: hand over between a cost and its gradient)
      if (need) {
        for (int k = 0; k < 3; k++) g[k] = -ba[2 + k];
        sm_grad(hs, g);
      }
      const bool dev_done = ((int)ba[ext + 2] & 1) != 0;
      if (dev_done != sm_done(hs) || (!dev_done && !same_bits(ba + ext + 3, sm_point(hs), 3))) { diverged = true; consumed++; break; }
      consumed++;
      if (c->chain_test == 1 && consumed == 3 && !sm_done(hs)) { diverged = true; break; }  // (test hook)
      continue;
    }
    // ---- replay: the cost stage
    const int ext_a = t.nout_a - kChainExtra;
    const bool dev_need = ba[ext_a] != 0.0;
    const bool need = sm_cost(hs, -ba[0]);
    if (need != dev_need) { diverged = true; break; }
    if (c->chain_test == 2 && need && consumed == 2) { diverged = true; break; }  // (test hook: hand over between a cost and its gradient)
    if (!need) {
      const bool dev_done = ba[ext_a + 2] != 0.0;
      if (dev_done != sm_done(hs) || (!dev_done && !same_bits(ba + ext_a + 3, sm_point(hs), 3))) { diverged = true; break; }
      consumed++;
      continue;
    }
    // ---- the gradient stage
    rc = wait_block(c, bg, t.g, t.nout_g);
    if (rc) break;
    for (int k = 0; k < 3; k++) g[k] = -bg[2 + k];
    sm_grad(hs, g);
    const int ext_g = t.nout_g - kChainExtra;
    const bool dev_done = bg[ext_g + 2] != 0.0;
    if (dev_done != sm_done(hs) || (!dev_done && !same_bits(bg + ext_g + 3, sm_point(hs), 3))) { diverged = true; consumed++; break; }
    consumed++;
    if (c->chain_test == 1 && consumed == 3 && !sm_done(hs)) { diverged = true; break; }  // (test hook: hand over between two points)
  }
  c->chain_active = false;
  // ---- leave the context in a state the ordinary evaluations understand.  Slots queued beyond the last one consumed either
  // return at once (the device's machine finished where the host's did) or must be stopped (divergence / error).
  if (rc || diverged || unsupported) {
    const int one = 1;
    (void)hipMemcpy(&c->d_chain->done, &one, sizeof(int), hipMemcpyHostToDevice);  // (null stream: overtakes the queued slots)
    (void)hipStreamSynchronize(c->stream);
    if (c->d_tail_counters) (void)hipMemsetAsync(c->d_tail_counters, 0, kTailCounterWords * sizeof(unsigned), c->stream);
    if (c->d_gacc) (void)hipMemsetAsync(c->d_gacc, 0, (size_t)kTailShards * kGaccStride * sizeof(double), c->stream);
    // The stop word lands in the middle of whatever slot is running: of a fused slot, some chunk workgroups may have arrived on tile
    // counters whose tile workgroups then left at once (counts never taken back), or tile workgroups may be waiting for chunks that left
    // (they give up after kFuseTimeoutTicks and flag the fallback word nobody reads any more).  Counters and word start from zero again.
    (void)fuse_incomplete;
    if (c->d_fnbr_cnt) (void)hipMemsetAsync(c->d_fnbr_cnt, 0, c->fcnt_cap * sizeof(unsigned), c->stream);
    if (c->d_fallback) (void)hipMemsetAsync(c->d_fallback, 0, sizeof(unsigned), c->stream);
    if (diverged) c->chain_takeovers++;
  }
  c->x_valid = false;
  c->jt_valid = false;
  if (!rc && !diverged && !unsupported && sm_done(hs) && all_self_gating && consumed >= 1 && c->pingpong_planes > 0 && c->d_accum_alt) {
    // Normal end: slots [0, consumed) ran, the ones queued behind them return on their first instruction and touch nothing.
    // So the planes of slot consumed-1 are the only dirty ones (its image pass cleared the partner), the moment rows it did not
    // add to are clear, accumulator rows and tickets are zero -- exactly what the next solve's first slot needs.  The host's
    // bookkeeping went on through the skipped slots: put it back on the buffer the last EXECUTED slot wrote.
    if ((queued - consumed) & 1) {
      std::swap(c->d_accum, c->d_accum_alt);
      std::swap(c->accum_cap, c->accum_alt_cap);
      std::swap(c->d_tflags, c->d_tflags_alt);
      std::swap(c->accum_flagged, c->alt_flagged);
    }
    c->accum_clean = false;
    c->alt_clean = true;
    c->chain_seq += (unsigned)consumed;
    c->chain_warm = true;
  } else {
    c->accum_clean = false;  // which of the two ping-pong buffers the last executed slot used is not known: both are cleared
    c->alt_clean = false;    // by the next evaluation's memset
  }
  c->accumulated = false;
  c->gated_pending = false;
  c->gate_mode = 0;
  if (rc) return rc;
  *completed = sm_done(hs) && !diverged;
  return CMX_OK;
}
