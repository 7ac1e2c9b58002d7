// cmx_pipeline.cpp -- one evaluation on the context's stream: accumulation buffers (ping-pong), destination-tile sort,
// image passes, gather, finalize and the completion ticket.  Shared by the front end and the back end.
#include <cstdlib>
#include "cmx_context.hpp"

// Fast path: swap to the partner buffer if it is known clean, otherwise clear the current one.  After this call
// c->d_accum is all-zero over `nplanes` planes and c->pingpong_planes tells the image pass to clear the partner.
int begin_accum(cmx_ctx *c, int nplanes, size_t np, bool fast) {
  c->pingpong_planes = 0;
  if (c->acc_dirty) {  // a split evaluation added to the accumulator rows and never reached its finalize (an error between the
                       // two phases): every later gradient would carry those sums -- the buffers are all-zero between launches
    if (c->d_gacc) HIP_TRY(c, hipMemsetAsync(c->d_gacc, 0, (size_t)kTailShards * kGaccStride * sizeof(double), c->stream));
    if (c->d_tail_counters) HIP_TRY(c, hipMemsetAsync(c->d_tail_counters, 0, kTailCounterWords * sizeof(unsigned), c->stream));
    c->acc_dirty = false;
  }
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
  const bool counting = count_sort_ok(ntiles + 1);
  if ((size_t)n > c->bin_cap || !c->d_keys) {
    uint32_t **ptrs[6] = {&c->d_keys, &c->d_sxy, &c->d_sbatch, &c->d_keys_s, &c->d_idx, &c->d_idx_s};
    for (int k = 0; k < 6; k++) {
      uint32_t **p = ptrs[k];
      if (*p) HIP_TRY(c, hipFree(*p));
      *p = nullptr;
      if (k >= 3 && counting) continue;  // the (key, index) pair buffers belong to the radix-sort fallback only
      HIP_TRY(c, hipMalloc((void **)p, (size_t)(n > 0 ? n : 1) * sizeof(uint32_t)));
    }
    c->bin_cap = (size_t)(n > 0 ? n : 1);
  }
  if (!counting && !c->d_keys_s) {  // a context that switched to a huge panorama after small ones
    uint32_t **ptrs[3] = {&c->d_keys_s, &c->d_idx, &c->d_idx_s};
    for (auto p : ptrs) HIP_TRY(c, hipMalloc((void **)p, c->bin_cap * sizeof(uint32_t)));
  }
  if (!c->d_fallback) {
    HIP_TRY(c, hipMalloc((void **)&c->d_fallback, sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(c->d_fallback, 0, sizeof(unsigned), c->stream));
  }
  rc = ensure(c, c->d_tile_start, c->tile_start_cap, (size_t)ntiles + 2);
  if (rc) return rc;
  // chunk size: hot tiles are split so that the chunks fill the chip in ONE resident round; big chunks amortise the window
  // zeroing and flush.  Back end: 135 VGPRs = 3 workgroups per CU, so n / 768 (swept at 5M events, splat us: 512 -> 50,
  // 640 -> 51, 768 -> 44, 1152 -> 50, 1536 -> 48, 3072 -> 53: anything that needs a second round loses more than it gains in
  // latency hiding; 128 VGPRs + 4 per CU and one event per thread + 4 per CU: no gain, 72 vs 81 us per cost evaluation).
  // Front end (50 VGPRs, LDS-bound): with 256-thread workgroups n / 512 (1M events, splat us: 256 -> 10.0, 384 -> 8.4, 512 -> 8.2,
  // 640 -> 8.5, 768 -> 8.8); round 5: 512-thread workgroups x chunks of n / 256 (profiles/r05_fe_shape.txt: 7.9 us; 1024 x n / 256: 7.7
  // but no better as an evaluation).
#ifndef CMX_FE_CHUNK_DIV
#define CMX_FE_CHUNK_DIV 256
#endif
  int M = n / (fe ? CMX_FE_CHUNK_DIV : 768);
  M = M < 1536 ? 1536 : (M > 32768 ? 32768 : M);  // floor swept on MI355X (1M events: 1536 -> 11.8 us, 1024 -> 15.3)
  M = (M + 255) / 256 * 256;
  // every tile contributes floor(len/M) full chunks and at most one remainder: an upper bound known on the host
  const int max_chunks = (n / M) + ntiles + 2 + n / 256;  // (+ the sentinel's events in chunks of 256: bound for 'all events rejected')
  rc = ensure(c, c->d_chunks, c->chunks_cap, (size_t)max_chunks);
  if (rc) return rc;
  if (!c->d_nchunks) HIP_TRY(c, hipMalloc((void **)&c->d_nchunks, sizeof(int)));
  if (!c->h_nchunks) {
    HIP_TRY(c, hipHostMalloc((void **)&c->h_nchunks, sizeof(unsigned long long), hipHostMallocMapped));
    *c->h_nchunks = 0ull;
    HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_nchunks_host, c->h_nchunks, 0));
  }
  c->binning_id++;  // a build_chunks of an earlier binning that was never collected may still store its (id, count): ignored
  if (c->binning_id == 0) c->binning_id = 1;
  if (n > 0) {
    if (counting) {
      // counting sort: keys + per-slice histograms, column prefixes, bin scan, scatter (cmx_binning.hip)
      rc = ensure(c, c->d_hist, c->hist_cap, count_sort_scratch_ints(n, ntiles + 1));
      if (rc) return rc;
      c->streams_valid = false;
      if (c->d_lut2) {  // the sorted order also carries every event's bearing (front end: and its dt)
        rc = ensure(c, c->d_sb, c->sb_cap, (size_t)2 * n);
        if (rc) return rc;
        if (fe) {
          rc = ensure(c, c->d_sdt, c->sdt_cap, (size_t)n);
          if (rc) return rc;
        }
        c->streams_valid = true;
      }
      launch_count_sort(fe, be, tiles_x, ntiles / planes_per_tile, c->d_xy, c->per_batch, n, c->d_keys, c->d_hist,
                        c->d_tile_start, c->d_sxy, c->d_sbatch, c->streams_valid ? c->d_sb : nullptr,
                        c->streams_valid ? c->d_sdt : nullptr, c->stream);
    } else {
      c->streams_valid = false;
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
    }
    // the chunk table is built where the offsets are: no read-back, no host loop, no synchronisation
    FusedTables ft{};
    c->fused_bin_id = 0;
    if (fe && c->fused_image && fused_tables_ok(ntiles, planes_per_tile)) {  // tables of the fused splat + image pass (FusedArgs)
      rc = ensure(c, c->d_fnbr_expected, c->fnbr_cap, (size_t)ntiles * kFuseStrips);
      if (rc) return rc;
      rc = ensure(c, c->d_fnbr_cnt, c->fcnt_cap, (size_t)ntiles * kFuseStrips * kFuseCntStride);
      if (rc) return rc;
      rc = ensure(c, c->d_fpartials, c->fpartials_cap, (size_t)2 * ntiles * kFuseStrips);
      if (rc) return rc;
      if (!c->d_ftiles_done) {
        HIP_TRY(c, hipMalloc((void **)&c->d_ftiles_done, sizeof(unsigned)));
        HIP_TRY(c, hipMalloc((void **)&c->d_fn_active, sizeof(int)));
      }
      if ((size_t)ntiles * kFuseStrips * kFuseCntStride > c->fdone_cap || !c->d_ftile_done) {
        rc = ensure(c, c->d_ftile_done, c->fdone_cap, (size_t)ntiles * kFuseStrips * kFuseCntStride);
        if (rc) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_ftile_done, 0, c->fdone_cap * sizeof(unsigned), c->stream));  // (no launch has stamp 0)
      }
      ft.n_active = c->d_fn_active;
      ft.tiles_done = c->d_ftiles_done;
      ft.tiles_y = tiles_y;
      ft.nbr_expected = c->d_fnbr_expected;
      ft.nbr_cnt = c->d_fnbr_cnt;
      ft.partials = c->d_fpartials;
      c->fused_bin_id = c->binning_id;
      c->fused_tiles_x = tiles_x;
      c->fused_tiles_y = tiles_y;
    }
    launch_build_chunks(c->d_tile_start, ntiles, planes_per_tile, tiles_x, kBinMargin, M, c->d_chunks, c->d_nchunks, c->d_nchunks_host,
                        c->binning_id, c->stream, ft.nbr_expected ? &ft : nullptr);
    HIP_TRY(c, hipGetLastError());
    c->nchunks = max_chunks;
    c->nchunks_exact = false;
  } else {
    HIP_TRY(c, hipMemsetAsync(c->d_nchunks, 0, sizeof(int), c->stream));
    c->nchunks = 0;
    c->nchunks_exact = true;
  }
  c->bin_valid = true;
  c->force_rebin = false;
  c->rebin_count++;
  c->last_fallback_frac = 0;
  return CMX_OK;
}

int ensure_fixed(cmx_ctx *c, size_t n) {
  if (n <= c->fixed_cap && c->d_fixed) return CMX_OK;
  int rc = ensure(c, c->d_fixed, c->fixed_cap, n);
  if (rc) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_fixed, 0, c->fixed_cap * sizeof(unsigned long long), c->stream));
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
  b.sort_tiles_x = (c->imgW + kBinTile - 1) / kBinTile;
  b.sort_tiles_y = (c->imgH + kBinTile - 1) / kBinTile;
  if (c->streams_valid) { b.sb = c->d_sb; b.sdt = c->kind == KIND_FE ? c->d_sdt : nullptr; }
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
  if (c->chain_active && !c->chain_first) {  // device-driven solve: omega and the end-of-solve flag live in device memory
    a.w_dev = c->d_chain->x_req;
    a.skip = &c->d_chain->done;
  }
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

// every evaluation ends in exactly one finalize -- its own launch (here) or the tail of the evaluation's last kernel
// (arm_tail) -- which carries the ticket sync_and_collect() waits for
void issue_finalize(cmx_ctx *c, FinalizeArgs &f, bool with_reduce) {
  f.ticket = ++c->ticket_issued;
  c->ticket_nout = 2 + (f.P > f.gP ? f.P : f.gP) + (f.chain.sm ? kChainExtra : 0);
  Span sp(c, CMX_T_FINAL, /*exact=*/true);
  if (with_reduce) launch_finalize(f, c->stream, sp.t0(), sp.t1());
  else launch_finalize_only(f, c->stream, sp.t0(), sp.t1());
}

// Tail finalize (CMX_OPT_TAIL_FINALIZE): hand the finalize to the launch that is about to be issued.  Only the forms whose
// inputs are per-workgroup partial tables of that very launch or results of earlier launches qualify (f.direct, no
// reduce_partials pass in between).
bool arm_tail(cmx_ctx *c, FinalizeArgs &f, TailArgs &tail, bool gated) {
  tail.counters = nullptr;
  if (!c->tail_finalize || !c->d_tail_counters || !f.direct || f.measure == 2) return false;
  // back end, gradient evaluations: read as a 42-column x 196-row table by one 256-thread workgroup the tail is 1.5-2 us
  // SLOWER than the finalize launch (measured, config 3).  Outside CMX_OPT_DETERMINISTIC the per-batch pass therefore adds
  // its column sums to kTailShards rows of accumulators with device-scope fp64 atomics (the order of the adds varies run
  // to run -- so does the vote image in that mode) and the tail reads 8 x 42 values.  Option value 2 forces the table form.
  if (c->kind == KIND_BE && f.gP > 0 && c->tail_finalize < 2) {
    if (c->deterministic || !c->d_gacc || 2 * f.gP > kGaccStride) return false;
    f.gacc = c->d_gacc;
    f.gacc_stride = kGaccStride;
  }
  // front end: the same accumulators instead of the ~1000 x 6 table (gather + tail 14.9 -> 13.2 us); value 2 keeps the table
  if (c->kind == KIND_FE && f.gP > 0 && c->tail_finalize < 2 && !c->deterministic && c->d_gacc) {
    f.gacc = c->d_gacc;
    f.gacc_stride = kGaccStride;
  }
  if (gated) {  // the gated pass reports to the second result block with its own ticket sequence
    f.ticket = ++c->ticket2_issued;
    c->ticket2_nout = 2 + (f.P > f.gP ? f.P : f.gP) + (f.chain.sm ? kChainExtra : 0);
  } else {
    f.ticket = ++c->ticket_issued;
    c->ticket_nout = 2 + (f.P > f.gP ? f.P : f.gP) + (f.chain.sm ? kChainExtra : 0);
  }
  tail.counters = c->d_tail_counters;
  tail.fin = f;
  // CMX_OPT_TAIL_FINALIZE 3 (A/B, measured without gain): front-end gradient evaluations through accumulator rows let workgroup 0 poll
  // sharded arrival counts instead of finding the last arriver through two levels of returning tickets
  tail.poll = (c->kind == KIND_FE && f.gP > 0 && f.gacc && c->tail_poll && !f.chain.sm) ? 1 : 0;
  return true;
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
  // The order of the list is the order finalize sums the tiles' moment rows in.  CMX_OPT_DETERMINISTIC: no list (which
  // tiles are listed -- dirty ones included -- depends on the history, and so would the grouping of the sum).  Sharded
  // evaluations: every rank must obtain bit-identical numbers or the replicated optimiser drivers stop taking the same
  // decisions; the ranks share flags and history, so a list built in TILE ORDER by one workgroup is the same on all of
  // them (the atomically compacted one is not).
  if (c->deterministic) return CMX_OK;
  const bool ordered = c->sharded();
  if ((size_t)a.nblk > c->tile_list_cap || !c->d_tile_list) {
    if (c->d_tile_list) HIP_TRY(c, hipFree(c->d_tile_list));
    if (c->d_tile_count) HIP_TRY(c, hipFree(c->d_tile_count));
    c->d_tile_list = c->d_tile_count = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_tile_list, (2 * (size_t)a.nblk + 8) * sizeof(unsigned)));  // list + dense scratch of the ordered form
    HIP_TRY(c, hipMalloc((void **)&c->d_tile_count, 2 * sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(c->d_tile_count, 0, 2 * sizeof(unsigned), c->stream));
    c->tile_list_cap = (size_t)a.nblk;
    c->tile_count_sel = 0;
  }
  unsigned *cur = c->d_tile_count + c->tile_count_sel, *next = c->d_tile_count + (c->tile_count_sel ^ 1);
  c->tile_count_sel ^= 1;  // this pass counts in `cur` and zeroes `next` for the pass after it
  launch_tile_list(a, reach, c->d_tile_list, cur, next, ordered, c->stream);
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
    Span sp(c, CMX_T_IMAGE, /*exact=*/true);
    launch_image_moments(a, c->stream, sp.t0(), sp.t1());
    launch_sobel_moments(sa, c->stream);
    FinalizeArgs f{};
    f.P = 0; f.nblk = a.nblk; f.measure = 2; f.npix = (double)np;
    f.partials = c->d_partials; f.sums = c->d_sums; f.result = result_ptr(c);
    f.direct = 1;
    f.gpartials = c->d_gpartials; f.gblocks = sa.nblk; f.gP = P;
    f.fallback = c->d_fallback;
    issue_finalize(c, f, false);
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  FinalizeArgs f{};
  f.P = P;
  f.nblk = a.nblk;
  f.measure = c->measure;
  f.npix = (double)np;
  f.partials = c->d_partials;
  f.sums = c->d_sums;
  f.result = result_ptr(c);
  f.fallback = c->d_fallback;
  bool tailed = false;
  {
    Span sp(c, CMX_T_IMAGE, /*exact=*/true);
    rc = maybe_tile_list(c, a, c->radius);
    if (rc) return rc;
    if (P == 0 && (a.nblk <= 2048 || a.tile_list)) {
      f.direct = 1;
      f.nvalid = a.tile_count;  // list path: partial rows are compact, one entry per listed tile
      if (!out_blur0 && !out_blurd) tailed = arm_tail(c, f, a.tail);  // the last-arriving workgroup finalizes
    }
    launch_image_moments(a, c->stream, sp.t0(), sp.t1());
  }
  if (!tailed) issue_finalize(c, f, /*with_reduce=*/!f.direct);
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}

// A cost-only evaluation that a gradient evaluation at the same parameters is likely to follow (the optimiser's f-then-df
// pattern, CMX_OPT_REUSE_IMAGE): run the fused adjoint image pass instead of the moments-only pass, so that the df finds
// Jt and the moment rows ready and launches its gather straight away (phase 3 of run_adjoint; +3..4 us per f, -12 us per df).
bool speculative_jt_ok(const cmx_ctx *c) {
  return c->reuse_image && adjoint_ok(c) && !c->accum_external && !c->sharded() && c->last_P == 0 && c->n_packed > 0;
}

// adjoint gradient: fused image pass (B = G*A with its moments, Jt = G^T B^) -> gather over the events (S1, and S2 for
// the votes next to the border) -> finalize: contrast from the moments, grad = (2/N)(S1 - mu*S2).
// phase 0 = everything; 1 = up to the per-rank partial sums (d_gsum, 2P doubles); 2 = finalize from d_gsum.
// (3: image pass + cost-only finalize, Jt kept; 4: the gated gradient pass behind it.)
int run_adjoint(cmx_ctx *c, int P, int phase) {
  const int W = c->imgW, H = c->imgH;
  const size_t np = (size_t)W * H;
  // image pass already done for these very planes (phase 2: by finish_begin; phase 0 after a speculative cost-only pass)
  // phase 4: the gated gradient pass queued behind a cost-only evaluation (phase 3) of the same point -- gather + tail finalize
  // into the second result block, running only if that evaluation's finalize opens the gate (cmx_hint_next_df)
  const bool have_image = phase == 2 || ((phase == 0 || phase == 4) && c->jt_valid);
  if (phase == 4 && !have_image) return CMX_OK;
  float *jt_before = c->d_itilde;
  int rc = ensure(c, c->d_itilde, c->itilde_cap, np);
  if (rc) return rc;
  if (c->d_itilde != jt_before)  // tiles the image pass skips keep whatever they held: make that finite from the start
    HIP_TRY(c, hipMemsetAsync(c->d_itilde, 0, c->itilde_cap * sizeof(float), c->stream));
  ImgAdjArgs ia{};
  ImgArgs &a = ia.img;
  if (c->composite_image && c->radius >= 1 && c->d_Mx && c->d_My && c->Mx_radius == c->radius) { ia.Mx = c->d_Mx; ia.My = c->d_My; }
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
  if (!have_image && c->pingpong_planes > 0 && c->d_accum_alt && !c->alt_clean) {
    a.zero_ptr = c->d_accum_alt;
    a.zero_planes = c->pingpong_planes;
    c->alt_clean = true;
  }
  if (!have_image) {
    rc = attach_tiles(c, a, /*may_skip=*/true);
    if (rc) return rc;
  }
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
  if (!have_image) c->adj_fused = false;  // the separate image pass below writes d_partials
  const bool fused_rows = have_image && c->adj_fused && phase != 2;  // moment rows of a pass that ran inside the splat launch
  FinalizeArgs f{};
  f.P = 0;
  f.nblk = fused_rows ? c->fused_tiles_x * c->fused_tiles_y * kFuseStrips : a.nblk;
  f.measure = c->measure;
  f.npix = (double)np;
  f.partials = fused_rows ? c->d_fpartials : c->d_partials;
  f.sums = c->d_sums;
  f.result = phase == 4 ? c->d_result2 : result_ptr(c);
  if (c->chain_active) {  // device-driven solve: this finalize advances the machine; results go to the slot's blocks of the ring
    f.result = phase == 4 ? c->chain_block_g : c->chain_block_a;
    f.chain.sm = &c->d_chain->sm;
    f.chain.x_req = c->d_chain->x_req;
    f.chain.done = &c->d_chain->done;
    f.chain.stage = phase == 4 ? 1 : 0;
    a.skip = &c->d_chain->done;
  }
  if (!have_image) {  // large panoramas: compact work list (a pre-pass kernel; partial rows become compact too)
    rc = maybe_tile_list(c, a, 2 * c->radius);
    if (rc) return rc;
  }
  bool direct = a.nblk <= 2048 || a.tile_list;  // few entries: finalize sums the per-tile moments itself
  if (have_image) {  // the image pass ran earlier -- its partial rows have the shape decided there
    direct = c->adj_direct;
    a.tile_count = c->adj_tile_count;
  } else {
    c->adj_direct = direct;
    c->adj_tile_count = a.tile_count;
  }
  f.direct = direct ? 1 : 0;
  f.nvalid = a.tile_count;
  f.mu_free = 1;
  // split call with the evaluator's own communicator: the workgroups' sums go to the accumulator rows (FinalizeArgs::gacc),
  // the rows are all-reduced as they are (cmx_comm.cpp) and the finalize sums them -- no per-batch kernel, no
  // reduce_gpartials launch.  Callers that all-reduce cmx_grad_ptr() themselves keep the 2P-double buffer.
  // (rank-invariant condition: every rank must issue the same collective, also one whose shard is empty)
  const bool acc_split = phase != 0 && c->shard_acc && !c->deterministic && c->d_gacc && 2 * P <= kGaccStride && P > 0;
  if (phase == 0 || phase == 4) {  // single call: finalize sums the gather kernel's block partials itself
    f.gpartials = c->d_gpartials;
    f.gblocks = gb;
  } else if (acc_split) {
    f.gacc = c->d_gacc;
    f.gacc_stride = kGaccStride;
  } else {           // split call: finalize reads the (all-reduced) per-parameter sums
    f.gpartials = c->d_gsum;
    f.gblocks = 1;
  }
  f.gP = P;
  f.fallback = c->d_fallback;
  if (phase == 2) {
    issue_finalize(c, f, false);
    HIP_TRY(c, hipGetLastError());
    c->acc_dirty = false;  // that finalize stores the zeros back
    return CMX_OK;
  }
  if (phase == 1 && acc_split) c->acc_dirty = true;  // until phase 2's finalize has been queued (see begin_accum)
  if (phase == 3) {
    f.gP = 0;
    f.gpartials = nullptr;
    f.gblocks = 0;
    if (c->gate_arm) {  // this finalize decides whether the pass queued behind it runs
      f.gate_out = c->d_gate;
      f.gate_thr = c->gate_thr;
      f.gate_mode = c->gate_mode;
    }
  }
  bool image_tailed = false;
  if (!have_image) {
    Span sp(c, CMX_T_IMAGE, /*exact=*/true);
    // cost-only with Jt kept (phase 3): the three-phase image pass has the registers to carry the finalize as its tail
    // (the five-phase kernel did not: inlined, its taps spilled and the pass went 11 -> 20 us)
    if (phase == 3 && ia.Mx && c->radius == 4) image_tailed = arm_tail(c, f, a.tail);
    launch_image_adjoint(ia, c->stream, sp.t0(), sp.t1());
    if (!direct) launch_reduce_partials(f, c->stream);
  }
  if (phase == 3) {  // cost-only: contrast from the moment rows; Jt and the rows stay for a gradient call at the same point
    if (!image_tailed) issue_finalize(c, f, false);
    HIP_TRY(c, hipGetLastError());
    c->jt_valid = true;
    c->spec_images++;
    return CMX_OK;
  }
  if (have_image && phase == 0 && !c->fused_done) c->spec_hits++;
  const bool gated = phase == 4;
  bool tailed = false;  // the gather launch carries the finalize
  {
    // (a gated launch may return on its first instruction: it is not a sample of the gather's duration)
    Span sp(c, CMX_T_GATHER, /*exact=*/true, /*enable=*/!gated);
    if (c->kind == KIND_FE) {
      FeGatherArgs g{};
      g.ev = fe_args(c, c->last_x);
      g.itilde = c->d_itilde;
      g.gpartials = c->d_gpartials;
      g.cx = c->d_cx; g.cy = c->d_cy; g.r = c->radius;
      if (c->splat_mode == 1 && c->bin_valid && !c->deterministic) {  // tile order: the sorted arrays the LDS splat consumes
        // (deterministic mode: time order -- the order inside a tile depends on the scatter's atomics)
        g.sxy = c->d_sxy;
        g.sbatch = c->d_sbatch;
        if (c->streams_valid) { g.sb = c->d_sb; g.sdt = c->d_sdt; }
      }
      if (!g.sxy && c->d_lut2 && c->n_packed > 0) {  // time-ordered gather: the events' bearings as a stream, once per packet
        if (!c->tb_valid) {
          int rc2 = ensure(c, c->d_tb, c->tb_cap, (size_t)2 * c->n_packed);
          if (rc2) return rc2;
          launch_bearing_stream(c->d_xy, c->d_lut2, c->W, c->n_packed, c->d_tb, c->stream);
          c->tb_valid = true;
        }
        g.tb = c->d_tb;
      }
      if (c->n_packed > 0) {
        if (phase == 0 || gated) tailed = arm_tail(c, f, g.tail, gated);
        else if (acc_split) g.tail.fin = f;  // (no counters: accumulators without the tail)
        if (gated) {
          if (!tailed) return CMX_OK;  // no tail available in this configuration: no gated pass (the df will run as usual)
          g.gate = c->d_gate;
          c->gated_pending = true;
          c->gated_launches++;
        }
        launch_fe_gather(g, c->stream, sp.t0(), sp.t1());
      } else {
        HIP_TRY(c, hipMemsetAsync(c->d_gpartials, 0, (size_t)gb * P2 * sizeof(double), c->stream));
      }
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
      g.deterministic = c->deterministic ? 1 : 0;
      if (c->d_lut2 && slice_shift == 8 && c->n_packed > 0) {  // once per window: the events' bearings in time order
        const int rc2 = be_ensure_time_bearings(c);
        if (rc2) return rc2;
        g.tb = c->d_tb;
      }
      if (c->n_packed > 0 && P > 0) {
        if (phase == 0 || gated) tailed = arm_tail(c, f, g.tail, gated);
        else if (acc_split) g.tail.fin = f;  // (no counters: accumulators without the tail)
        g.fold = c->fold_batch ? 1 : 0;
        if (gated) {  // only the one-kernel form can be gated
          if (!tailed || !be_gather_folds(g)) return CMX_OK;
          g.gate = c->d_gate;
          c->gated_pending = true;
          c->gated_launches++;
        }
        if (be_gather_folds(g)) {  // one kernel: gather, per-batch pass and finalize
          launch_be_gather(g, c->nb, c->stream, sp.t0(), sp.t1(), nullptr, nullptr);
        } else {
          Span sb(c, CMX_T_BATCH, /*exact=*/true);
          launch_be_gather(g, c->nb, c->stream, sp.t0(), sp.t1(), sb.t0(), sb.t1());
        }
      } else {
        HIP_TRY(c, hipMemsetAsync(c->d_gpartials, 0, (size_t)gb * P2 * sizeof(double), c->stream));
      }
    }
    if (phase == 1 && !acc_split) launch_reduce_gpartials(c->d_gpartials, gb, 2 * P, c->d_gsum, c->stream);
  }
  if (phase == 1 || gated) {
    HIP_TRY(c, hipGetLastError());
    return CMX_OK;
  }
  if (!tailed) issue_finalize(c, f, false);
  HIP_TRY(c, hipGetLastError());
  return CMX_OK;
}

// spin on a completion ticket in a mapped result block; true once a consistent snapshot carrying `want` has been read
// (budget_us >= 0: give up after that long -- the caller then blocks in hipStreamSynchronize; -1: the 20 ms safety bound only)
bool spin_for_ticket(const double *h_block, unsigned long long want, int nout, int budget_us) {
  const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(h_block);
  const auto t0 = std::chrono::steady_clock::now();
  bool done = false;
  const double limit_ms = budget_us >= 0 ? budget_us * 1e-3 : 20.0;
  const unsigned check = budget_us >= 0 ? 63u : 1023u;
  for (unsigned spins = 0;; spins++) {
    if (w[kTicketSlot] == want) {  // ticket seen: accept only a consistent snapshot of the results
      unsigned long long x = w[kFallbackSlot];
      for (int k = 0; k < nout; k++) x ^= w[k];
      if ((x ^ (want * kTicketMix)) == w[kChecksumSlot]) { done = true; break; }
    }
    __builtin_ia32_pause();
    if ((spins & check) == check &&
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit_ms)
      break;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return done;
}

int sync_and_collect(cmx_ctx *c, bool ends_in_finalize) {
  // ends_in_finalize: the last thing queued on the stream is an evaluation's finalize kernel.  Wait for it through
  // its completion ticket in mapped host memory (a few microseconds earlier than the runtime reports the stream idle);
  // anything slower than the spin budget, and every caller that queued copies or other kernels after the finalize,
  // takes the ordinary stream synchronisation.
  bool done = false;
  if (ends_in_finalize && c->ticket_wait && c->ticket_issued) done = spin_for_ticket(c->h_result, c->ticket_issued, c->ticket_nout, c->spin_eval_us);
  if (!done) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (ends_in_finalize && c->ticket_issued) {
      // the stream is idle: the finalize (its own launch, or the tail of the last kernel) must have delivered its ticket
      const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(c->h_result);
      if (w[kTicketSlot] != c->ticket_issued) {
        const unsigned long long got = w[kTicketSlot];
        if (c->d_tail_counters) (void)hipMemsetAsync(c->d_tail_counters, 0, kTailCounterWords * sizeof(unsigned), c->stream);
        if (c->d_gacc) (void)hipMemsetAsync(c->d_gacc, 0, (size_t)kTailShards * kGaccStride * sizeof(double), c->stream);
        return fail(c, CMX_ERR_HIP, "evaluation ended without its finalize step (ticket %llu, expected %llu)", got,
                    (unsigned long long)c->ticket_issued);
      }
    }
  }
  if (!c->nchunks_exact && c->bin_valid && c->d_nchunks) {  // once per binning: launch exactly the chunks that exist
    // the kernel that built the table also stored its length in mapped host memory; a kernel queued behind it has delivered
    // its completion ticket by now, so that store has landed (no synchronous read-back: ~10 us per packet)
    int nch = -1;
    if (c->h_nchunks) {
      const unsigned long long w = *reinterpret_cast<volatile unsigned long long *>(c->h_nchunks);
      if ((unsigned)(w >> 32) == c->binning_id) nch = (int)(unsigned)(w & 0xffffffffull);
    }
    if (nch < 0 && hipMemcpy(&nch, c->d_nchunks, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) nch = -1;
    if (nch >= 0 && nch <= c->nchunks) c->nchunks = nch;
    c->nchunks_exact = true;
  }
  if (c->fallback_pending && c->n_packed > 0) {
    c->last_fallback_frac = fallback_count(c->h_result[kFallbackSlot]) / (double)c->n_packed;
    c->last_fallback_flags = fallback_flags(c->h_result[kFallbackSlot]);
  }
  c->fallback_pending = false;
  // timing spans are resolved lazily (cmx_timing_get) so that timed evaluations wait exactly like untimed ones
  if (c->spans.size() > 4096) {
    if (done) HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_spans(c);
  }
  return CMX_OK;
}

bool can_reuse(const cmx_ctx *c, const double *x, int n, bool want_grad) {
  if (!want_grad || !c->reuse_image || !c->have_data || !c->accumulated || !c->x_valid) return false;
  if (!adjoint_ok(c) || c->accum_external) return false;
  return memcmp(x, c->last_x, sizeof(double) * (size_t)n) == 0;
}

// ---- gated gradient pass (cmx_hint_next_df).  The FR-CG line search asks for the gradient at the point of a cost-only
// evaluation exactly when that evaluation's cost passes a test it knows beforehand (f < fa on the trial step, !(f >= fa)
// in the bracketing loop, f <= fb in Brent's loop).  With the test handed over in front of the cost-only evaluation, its
// finalize writes the outcome to d_gate and the gradient pass -- queued behind it right away, reporting to the second result
// block -- runs or returns at once.  The df call that follows finds its result in flight instead of starting a launch:
// one host-to-GPU turnaround (~6 us) less per accepted point.
int finish_cost_only_speculative(cmx_ctx *c, int P) {
  const bool want_gate = c->gated_df && c->gate_mode != 0 && !c->sharded() && !c->accum_external && c->ticket_wait;
  c->gated_pending = false;
  c->gate_arm = want_gate;
  int rc = run_adjoint(c, P, /*phase=*/3);
  c->gate_arm = false;
  const int mode = c->gate_mode;
  const double thr = c->gate_thr;
  c->gate_mode = 0;  // the hint is for ONE evaluation
  if (rc) return rc;
  if (want_gate) {
    rc = run_adjoint(c, P, /*phase=*/4);  // sets gated_pending when the pass could be queued
    if (rc) return rc;
  }
  rc = sync_and_collect(c, true);  // waits for the cost-only finalize's ticket, not for the pass queued behind it
  if (rc) { c->gated_pending = false; return rc; }
  if (c->gated_pending) c->gated_fired = gate_condition(c->h_result[0], thr, mode) != 0;  // the device evaluated the same expression
  return CMX_OK;
}

// gradient evaluation at the point of the last cost-only evaluation: served by the gated pass if one is in flight
int collect_gated(cmx_ctx *c, int P, double *contrast, double *grad, bool *served) {
  *served = false;
  if (!c->gated_pending) return CMX_OK;
  c->gated_pending = false;
  if (!c->gated_fired) return CMX_OK;  // the gate stayed shut (the caller asks anyway): ordinary gradient pass
  bool done = spin_for_ticket(c->h_result2, c->ticket2_issued, c->ticket2_nout, c->spin_eval_us);
  if (!done) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (reinterpret_cast<const unsigned long long *>(c->h_result2)[kTicketSlot] != c->ticket2_issued)
      return CMX_OK;  // the stream is idle and the pass did not report (its gate stayed shut): ordinary gradient pass
  }
  *contrast = c->h_result2[0];
  for (int k = 0; k < P; k++) grad[k] = c->h_result2[2 + k];
  c->gated_hits++;
  c->spec_hits++;
  *served = true;
  return CMX_OK;
}

// ---- three-phase finish for sharded adjoint evaluations: begin (image, adjoint blur, gather -> partial gradient
// sums on the device), caller all-reduces cmx_grad_ptr(), end (finalize + read-back)
int finish_begin(cmx_ctx *c, int kind, int want_grad) {
  if (!c || c->kind != kind) return fail(c, CMX_ERR_STATE, "wrong context kind");
  if (!c->accumulated) return fail(c, CMX_ERR_STATE, "finish_begin without accumulate");
  int rc = bind_device(c);
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
int finish_end(cmx_ctx *c, int kind, double *contrast, double *grad) {
  if (!c || c->kind != kind) return fail(c, CMX_ERR_STATE, "wrong context kind");
  if (!c->finish_pending) return fail(c, CMX_ERR_STATE, "finish_end without finish_begin");
  if (!contrast) return fail(c, CMX_ERR_INVALID_ARG, "null contrast");
  int rc = bind_device(c);
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
int cmx_backend_finish_begin(cmx_ctx *c, int want_grad) { CMX_NOT_FOR_GROUPS(c, "the split-phase interface"); return finish_begin(c, KIND_BE, want_grad); }
int cmx_backend_finish_end(cmx_ctx *c, double *contrast, double *grad) { CMX_NOT_FOR_GROUPS(c, "the split-phase interface"); return finish_end(c, KIND_BE, contrast, grad); }
void *cmx_grad_ptr(const cmx_ctx *c) { return c ? c->d_gsum : nullptr; }
size_t cmx_grad_count(const cmx_ctx *c) { return (c && c->finish_pending && c->pending_P > 0) ? (size_t)(2 * c->pending_P) : 0; }
int cmx_set_grad_buffer(cmx_ctx *c, void *device_ptr, size_t n_doubles) {
  if (!c) return CMX_ERR_INVALID_ARG;
  CMX_NOT_FOR_GROUPS(c, "a caller-owned gradient buffer");
  int rc = bind_device(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!c->gsum_external && c->d_gsum) HIP_TRY(c, hipFree(c->d_gsum));
  c->d_gsum = (double *)device_ptr;
  c->gsum_cap = device_ptr ? n_doubles : 0;
  c->gsum_external = device_ptr != nullptr;
  return CMX_OK;
}


// ---- several INDEPENDENT evaluations per call (cmx_*_eval_many): the launch chains of the m evaluations are queued back
// to back on the stream, every finalize writes to its own block of mapped host memory, and the host waits ONCE.  What a
// sequential caller pays per evaluation on top of the kernels -- ticket -> host -> next launch, ~3 us -- disappears, and
// the GPU never idles between evaluations.  For candidate lists (multi-start, grid initialisation, finite-difference
// checks); a line search cannot use it (each trial point depends on the previous result).
static int eval_many(cmx_ctx *c, int kind, int m, const double *xs, double *contrasts, double *grads) {
  if (!c || c->kind != kind) return fail(c, CMX_ERR_STATE, "wrong context kind");
  if (!c->have_data) return fail(c, CMX_ERR_STATE, "no packet / window handed over");
  if (m < 0 || (m > 0 && (!xs || !contrasts))) return fail(c, CMX_ERR_INVALID_ARG, "bad arguments");
  if (c->sharded()) return fail(c, CMX_ERR_STATE, "eval_many is not available with a communicator attached");
  int rc = bind_device(c);
  if (rc) return rc;
  const int n = kind == KIND_FE ? 3 : 3 * (c->K - c->num_fixed);
  constexpr int kBlock = 4096;  // doubles per evaluation: the layout of the one-evaluation result buffer (slots at its tail)
  if ((size_t)m > c->many_cap) {
    if (c->h_many) HIP_TRY(c, hipHostFree(c->h_many));
    c->h_many = c->d_many = nullptr;
    c->many_cap = 0;
    HIP_TRY(c, hipHostMalloc((void **)&c->h_many, (size_t)m * kBlock * sizeof(double), hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_many, c->h_many, 0));
    c->many_cap = (size_t)m;
  }
  const int reuse = c->reuse_image;
  c->reuse_image = 0;  // every evaluation of the list is a full one; no speculative work for a follow-up call
  c->gate_mode = 0;    // (a hint belongs to the ONE cost-only evaluation it was given for)
  c->gated_pending = false;
  for (int i = 0; i < m && rc == CMX_OK; i++) {
    const double *x = xs + (size_t)i * n;
    rc = kind == KIND_FE ? cmx_frontend_accumulate(c, x, grads != nullptr) : cmx_backend_accumulate(c, x, grads != nullptr);
    if (rc) break;
    if (kind == KIND_BE) {
      rc = be_first_iter(c);
      if (rc) break;
    }
    c->result_override = c->d_many + (size_t)i * kBlock;
    if (grads && c->last_adjoint) rc = run_adjoint(c, n);
    else rc = run_image_and_finalize(c, grads ? n : 0, nullptr, nullptr);
    c->result_override = nullptr;
  }
  c->reuse_image = reuse;
  c->x_valid = false;  // (the resident image belongs to the last point of the list: not worth tracking)
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (rc) return rc;
  HIP_TRY(c, e);
  for (int i = 0; i < m; i++) {
    const double *r = c->h_many + (size_t)i * kBlock;
    contrasts[i] = r[0];
    if (grads) for (int k = 0; k < n; k++) grads[(size_t)i * n + k] = r[2 + k];
  }
  if (m > 0 && c->n_packed > 0 && c->last_used_lds)
    c->last_fallback_frac = fallback_count(c->h_many[(size_t)(m - 1) * kBlock + kFallbackSlot]) / (double)c->n_packed;
  c->fallback_pending = false;
  return CMX_OK;
}
int cmx_frontend_eval_many(cmx_ctx *c, int m, const double *omegas, double *contrasts, double *grads) {
  UrgentScope urgent(c);
  return eval_many(c, KIND_FE, m, omegas, contrasts, grads);
}
int cmx_backend_eval_many(cmx_ctx *c, int m, const double *drotvs, double *contrasts, double *grads) {
  UrgentScope urgent(c);
  return eval_many(c, KIND_BE, m, drotvs, contrasts, grads);
}

// ---- m DEPENDENT-style evaluations per call (cmx_*_eval_each): exactly m calls of cmx_*_eval, each waited for before the
// next is issued, without returning to the caller in between -- what an optimiser loop written in C/C++ does.  For hosts
// whose own loop is expensive per call (an interpreter): the evaluation sequence of a line search cannot be known in
// advance, but replaying a recorded one, or timing the evaluator without the caller's overhead, can use it.
int cmx_frontend_eval_each(cmx_ctx *c, int m, const double *omegas, double *contrasts, double *grads) {
  UrgentScope urgent(c);
  if (!c || m < 0 || (m > 0 && (!omegas || !contrasts))) return c ? fail(c, CMX_ERR_INVALID_ARG, "bad arguments") : CMX_ERR_INVALID_ARG;
  for (int i = 0; i < m; i++) {
    const int rc = cmx_frontend_eval(c, omegas + 3 * (size_t)i, contrasts + i, grads ? grads + 3 * (size_t)i : nullptr);
    if (rc) return rc;
  }
  return CMX_OK;
}
int cmx_backend_eval_each(cmx_ctx *c, int m, const double *drotvs, double *contrasts, double *grads) {
  UrgentScope urgent(c);
  if (!c || m < 0 || (m > 0 && (!drotvs || !contrasts))) return c ? fail(c, CMX_ERR_INVALID_ARG, "bad arguments") : CMX_ERR_INVALID_ARG;
  const size_t P = (size_t)(c->order > 0 ? 3 * (c->K - c->num_fixed) : 0);
  for (int i = 0; i < m; i++) {
    const int rc = cmx_backend_eval(c, drotvs + P * i, contrasts + i, grads ? grads + P * i : nullptr);
    if (rc) return rc;
  }
  return CMX_OK;
}
