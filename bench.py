#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X event-warping path (BASELINE.json metric).

A "step" is one full cost+gradient evaluation (what local_contrast_fdf / global_contrast_fdf do once: warp + splat every
event, blur, contrast and its analytic gradient) over one batch of synthetic input resident in HBM.  The timed loop
cycles the parameters through points of a RECORDED solve trajectory, so the destination-tile sort of the first
evaluation sees the drift a real solve produces (votes leaving their LDS windows, re-sorts).

  N = 1 (default) : `value` = BASELINE config 2 -- 1M synthetic events, 640x480 IWE, front-end CMax, one MI355X.
                    Nested in the same JSON line: "backend" = BASELINE config 3 (5M events, cubic 10-knot spline,
                    1024x1024 panorama) measured the same way.
  N > 1 (torchrun): `value` = BASELINE config 4 -- back-end BA window, 5M events PER GPU (40M on 8), sharded by contiguous
                    event-batch ranges, one process per GPU, RCCL all-reduce of the partial panoramas between splat and
                    blur and of the 2P partial gradient sums after the gather (weak scaling).  Nested: "config5" = 20M
                    events / 8 per GPU, 1280x720 sensor, 4096x2048 map.  --workload frontend keeps round 1's front-end
                    variant (N x 1M events over one packet).

  N > 1 (plain `python bench.py --gpus N`, no launcher): the same config 4 through ONE handle -- the one-process group
                    (cmx_backend_create_group over devices 0..N-1, the form the reference's single back-end thread can use) is
                    the headline, and the process-per-GPU form is self-spawned (torch.distributed.run) as a nested leg.  Fewer
                    devices than N: what is there runs, `n_gpus` says how many, an `error` field says why; exit status 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|frontend|backend] [--no-backend] [--no-cpu-baseline]
Rank 0 prints ONE JSON line on stdout, at most LINE_BUDGET bytes: the contract's keys, `roofline`, `cpu_baseline`, `summary`.
Everything else measured in the run (per-kernel table, per-packet / per-window pipelines, launch-default shapes, front end beside
back end, group, large launch, concurrent contexts, the nested back-end leg in full) goes to bench_detail.json (--detail-out) and,
as one line, to stderr.  Every number is measured in this run except `roofline.traffic` / `kernels[].pmc_bytes`, which are
read from profiles/pmc_traffic.json (rocprofv3 --pmc passes of this same command, see profiles/README.md).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# fp64 vector issue: 256 CUs x 4 SIMDs x 2.4 GHz, one wave64 fp64 instruction per 4 clocks per SIMD (78.6 TFLOP/s fp64 vector = 2 flop x
# 64 lanes x this) = 614.4 G wave-instructions/s = 39.3 T lane-instructions/s.  The back-end kernels' VALU instructions are a mix (fp64
# at 4 clocks, fp32 / integer at 2): pricing every one of them at the fp64 rate is the generous-to-the-hardware reading, frac <= 1.
VALU_FP64_PEAK_TLANE = 256 * 4 * 2.4e9 / 4 * 64 / 1e12
METRIC = "warped-events/sec/GPU (1M ev, 640x480 IWE) + CMax iters/sec"


# ---------------------------------------------------------------------------------------------- byte models
SETTLE_STEPS_MULTI = 400
SETTLE_S = 0.2  # untimed clock-settling evaluations in front of every workload's warm-up (see measure())


def byte_models(kind, order, n_events, npix, nb, P, adjoint, nnz_pixels, image_pixels=None):
    """Per kernel class and per launch: (algorithmic bytes, HBM-mandatory bytes).

    ALGORITHMIC = SURVEY.md section 8(d): per warped event 4 B coordinates + 24 B fp64 bearing + 4 px x (4 B read + 4 B
    write) per image voted into; the gather pass 4 + 24 + 4 px x 4 B; image passes W*H*4 B per plane touch.
    HBM-MANDATORY = the bytes that have to cross the memory interface at least once per launch given the data layout
    actually used: the coalesced per-event streams, one read / write of every plane a kernel
    consumes / produces, one 4-byte write per non-zero IWE pixel for the LDS-window flush.  Vote read-modify-writes live
    in LDS and table gathers hit L2, so they are not in it.  `frac` (mandatory bytes / time / peak) can therefore never exceed 1."""
    fe = kind == "frontend"
    ipx = npix if image_pixels is None else image_pixels  # pixels the image passes touch (back end: occupied tiles + filter reach)
    imgs = 1 if adjoint else (1 + (3 if fe else 3 * order))
    planes_in = 1 if fe else 2
    m = {}
    m["splat"] = (n_events * (4 + 24 + imgs * 32),
                  n_events * 24 + nnz_pixels * 4 + (0 if fe else nb * 72))
    #   front end: bearing 16 B + dt 8 B per event (streams in tile order); back end: packed event 4 B + batch index 4 B +
    #   bearing 16 B per event, the 72-byte rotation of every batch once
    if adjoint:
        m["gather"] = (n_events * (4 + 24 + 16),
                       n_events * (24 if fe else 20) + ipx * 4 + (0 if fe else nb * 72 + nb * 48))
        m["image"] = (npix * 4 * (planes_in + 1 + planes_in), ipx * 4 * (planes_in + 1 + planes_in))
        #   read the vote plane(s), write Jt, clear the ping-pong partner's plane(s)
    else:
        m["image"] = (npix * 4 * 6 * (1 + P), ipx * 4 * (planes_in + P))
    m["image_f"] = (npix * 4 * 2 * planes_in, ipx * 4 * 2 * planes_in)  # cost-only: read the plane(s), clear the partner
    if not fe:
        m["pose"] = (nb * (8 + 72 + 152), nb * (8 + 72 + 152))
        m["batch"] = (nb * (48 + 152), nb * (48 + 152))
    m["final"] = (0, 0)
    return m


def whole_eval_bytes_8d(kind, order, n_events, npix, P):
    """SURVEY.md section 8(d)'s whole-evaluation figure for one fdf of the REFERENCE's data flow: N * B_ev + planes * W*H*4*6."""
    per_ev = {"frontend": 156, "backend2": 252, "backend4": 444}["frontend" if kind == "frontend" else "backend%d" % order]
    return n_events * per_ev + (1 + P) * npix * 4 * 6


def csrc_hash():
    """sha256 over cmax_slam_amd/csrc (the sources libcmaxhip.so is built from); tools/pmc_to_json.py stamps the same hash into
    profiles/pmc_traffic.json."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cmax_slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def load_pmc():
    """(per-kernel PMC byte counts, note).  The counts are only reported for the build they were measured on: the file carries
    the source hash of cmax_slam_amd/csrc at measurement time; on a mismatch `traffic` / `pmc_bytes` are null."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}, "profiles/pmc_traffic.json missing"
    stamp = pmc.get("_stamp") or {}
    if stamp.get("src_sha256") != csrc_hash():
        return {}, ("profiles/pmc_traffic.json was measured on another build (source hash %s..., running %s...): not reported"
                    % (str(stamp.get("src_sha256"))[:12], csrc_hash()[:12]))
    return pmc, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build (%s, git %s)" % (pmc.get("_source"), str(stamp.get("git_head"))[:12])


# What the counters and the same-box A/B diagnostics say binds the dominant kernels (`bound` stays the roofline `frac` is computed
# against -- HBM, the only roofline this scatter / gather / stream path has; it is not what the launch waits for at these sizes).
LIMITED_BY = {
    ("frontend", "gather"): "launch + tail latency: ~4 us of launch latency + 4.7 us last-arriver finalize of 13.9 us, whatever the launch shape "
                            "(977, 489 or 245 workgroups: 13.6-13.8 us, profiles/r05_fe_shape.txt); the event loop runs 6 us = 3.9 TB/s "
                            "(profiles/r04_fe_gather_anatomy.txt)",
    ("backend", "gather"): "fp64 VALU issue (230 VALU instructions/event, pipes 0.57 busy) + the dependent steps of a wave iteration at 3 "
                           "waves/SIMD (half of a wave's life is a wait); 4 waves, LDS-held coefficients and interleaved polynomial chains "
                           "measured and rejected (profiles/r05_be_valu.txt)",
    ("backend", "splat"): "per-iteration dependent chain (stream loads -> pose gather -> projection -> LDS votes): VALU pipes 0.47 busy, HBM 0.60 "
                          "by the counters; more waves per SIMD buy nothing (profiles/r05_be_valu.txt, r04_be_splat_pose_traffic.txt)",
    ("frontend", "splat"): "launch latency of its shape (profiles/r02_splat_timeline.txt, r05_fe_shape.txt)",
}


# ---------------------------------------------------------------------------------------------- measurement core
class Runner:
    """One evaluator + its exchange path; knows how to run an fdf / cost-only step at a given parameter vector."""

    def __init__(self, ev, world, comm_mode, device, torch, dist, force=False):
        self.ev, self.world, self.torch, self.dist, self.device = ev, world, torch, dist, device
        self.comm_used = "none"
        self.sh = self.stream = None
        self.ev_sharded = world > 1 or force
        if world > 1 or force:  # force: the sharded code path with a 1-rank communicator (--force-sharded, a dry run)
            self._attach(comm_mode)

    def _attach(self, comm_mode):
        torch, dist, ev = self.torch, self.dist, self.ev
        from cmax_slam_amd.dist import ShardedEvaluator, attach_torch_accum
        rank = dist.get_rank()
        mode = comm_mode
        if mode == "native":
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=self.device)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(ev.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, src=0)
                ev.comm_attach(bytes(idt.cpu().numpy().tobytes()), rank, self.world)
                self.comm_used = "native RCCL communicator inside the evaluator"
            except Exception as e:  # keep the run alive: fall back to torch.distributed on evaluator-owned tensors
                self.comm_used = "torch.distributed (native attach failed: %s)" % e
                mode = "torch"
            ok = torch.tensor([1 if mode == "native" else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks must take the same exchange path
            if int(ok.item()) == 0 and mode == "native":
                ev.comm_detach()
                self.comm_used = "torch.distributed (native attach failed on another rank)"
                mode = "torch"
        if mode == "torch":
            accum, gsum, self.stream = attach_torch_accum(ev, self.device)
            self.sh = ShardedEvaluator(ev, accum, gsum)
            if self.comm_used == "none":
                self.comm_used = "torch.distributed.all_reduce (RCCL) in place on the evaluator's planes"

    def step(self, x, want_grad=True):
        if self.sh is not None:
            with self.torch.cuda.stream(self.stream):
                return self.sh.eval(x, want_grad)
        return self.ev.eval(x, want_grad)

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def record_trajectory(ev, x0, kind, params, max_points=8):
    """Evaluation points of one real FR-CG solve (the restated driver over THIS evaluator, python callbacks, untimed),
    thinned to <= max_points spread over the solve: first point, accepted iterates, trial points, last point."""
    from cmax_slam_amd import solver
    pts = []

    def fdf(x, wg):
        pts.append(np.array(x, dtype=np.float64))
        c, g = ev.eval(x, wg)
        return -c, (-g if wg else None)
    solver.frcg_minimize(fdf, x0, **params)
    uniq = []
    for p in pts:
        if not any(np.array_equal(p, q) for q in uniq):
            uniq.append(p)
    if len(uniq) > max_points:
        idx = np.unique(np.round(np.linspace(0, len(uniq) - 1, max_points)).astype(int))
        uniq = [uniq[i] for i in idx]
    return uniq


def measure(run, points, steps, warmup, kind, order, n_local, n_total, npix, nb, P, adjoint, pmc_prefix):
    """Warm-up, per-kernel calibration (HIP events carried by every kernel class), then the timed region (EXACTLY `steps`
    fdf evaluations between two fences, dominant kernel timed live on ~12 of them), then the cost-only loop."""
    ev = run.ev
    npts = len(points)
    # the GPU drops its clocks while the host works (set-up, the recorded solve's python callbacks, the previous workload's
    # CPU baseline): W evaluations of 50-160 us are over before they are back (tools/traj_points.py: the first ~10 ms after
    # an idle period run 8-10 % slow).  Untimed, like the warm-up: evaluations for SETTLE_S seconds of wall clock first.
    if run.world > 1:  # ranks must issue the same number of collectives: a fixed count instead of a wall-clock bound
        for i in range(SETTLE_STEPS_MULTI):
            run.step(points[i % npts], True)
    else:
        t_end = time.perf_counter() + SETTLE_S
        i = 0
        while time.perf_counter() < t_end:
            run.step(points[i % npts], True)
            i += 1
    for i in range(warmup):
        run.step(points[i % npts], True)
    # ---- calibration (untimed): every kernel class, fdf then cost-only
    ncal = max(8 * npts, 64)  # (16 evaluations gave per-kernel means that moved by 10 % from run to run)
    ev.timing_enable(True)
    ev.timing_get()
    for i in range(ncal):
        run.step(points[i % npts], True)
    run.torch.cuda.synchronize()
    cal = ev.timing_get()
    for i in range(ncal):
        run.step(points[i % npts], False)
    run.torch.cuda.synchronize()
    cal_f = ev.timing_get()
    kernel_ms = {k: v[0] / v[1] for k, v in cal.items() if v[1] and k not in ("zero",)}
    kernel_ms_f = {k: v[0] / v[1] for k, v in cal_f.items() if v[1] and k not in ("zero",)}
    per_eval = {k: v[1] / ncal for k, v in cal.items() if v[1]}
    compute = [k for k in kernel_ms if k != "comm"]
    dom = max(compute, key=lambda k: kernel_ms[k])
    # ---- timed region
    # The K timed steps are issued by ONE native call (cmx_*_eval_each): exactly K calls of cmx_*_eval, each waited for before
    # the next is issued -- the C++ loop of the reference's host (GSL inside a ROS node), without ~2 us of Python interpreter
    # per step between the evaluations.  (--comm torch keeps the Python loop: its exchange lives in Python.)
    native_loop = run.sh is None
    xs_timed = np.vstack([points[i % npts] for i in range(steps)])
    # Round 6: NO launch inside the timed region carries timing events.  A launch with hipExtLaunchKernelGGL start / stop signals costs
    # its step 4-16 us, i.e. 0.8-2.2 us on every step of a K = 20 region (tools/fixed_overhead.py): the headline is the K
    # plain evaluations and nothing else.  The dominant kernel's duration is measured in a sample pass right BEHIND the region (the same
    # points, the same resident state, every step timed) and cross-checked against the calibration pass in front of it.
    if native_loop:  # (untimed) the calibration's read-back left the GPU idle for a millisecond or two: a short timed region would
        ev.eval_each(xs_timed[:16], True)  # otherwise start on a device that has begun to clock down (K = 20: +1.5 us per step)
    ev.timing_enable(False)
    timed_call = ev.prepare_eval_each(xs_timed, True) if native_loop else None  # (buffers and pointers made in front of the region)
    run.fence()
    t0 = time.perf_counter()
    if native_loop:
        cs, gs = timed_call()
        c, g = float(cs[-1]), gs[-1].copy()
    else:
        for i in range(steps):
            c, g = run.step(points[i % npts], True)
    t_loop = time.perf_counter()  # (diagnostic split of the region: the K evaluations | the closing fence; `elapsed` is the whole of it)
    run.fence()
    elapsed = time.perf_counter() - t0
    region_split = {"loop_ms": (t_loop - t0) * 1e3, "closing_fence_ms": elapsed * 1e3 - (t_loop - t0) * 1e3}
    # ---- sample pass behind the region: the dominant kernel's live duration (HIP events carried by its dispatches)
    # (24 launches read 10-25 % above rocprofv3's average of the same kernel in the same loop -- 20.9 against 16.9 us, tools/live_vs_trace.py:
    #  the first launches that carry events pay for the events themselves and a mean of 24 keeps that; primed and over 200+ launches
    #  the two agree within 3 %: 17.1 against 16.7 us)
    nsample = max(25 * npts, 200)
    ev.timing_enable([dom])
    if native_loop:  # the same native loop as the region: a Python loop leaves gaps between the evaluations in which the device clocks
        ev.eval_each(np.vstack([points[i % npts] for i in range(2 * npts)]), True)  # (priming, discarded)
        ev.timing_get()
        ev.eval_each(np.vstack([points[i % npts] for i in range(nsample)]), True)  # down (sampled durations read 15-20 % long)
    else:
        for i in range(2 * npts):
            run.step(points[i % npts], True)
        ev.timing_get()
        for i in range(nsample):
            run.step(points[i % npts], True)
    run.fence()
    tim = ev.timing_get()
    ev.timing_enable(False)
    stats = ev.stats()
    for i in range(16):
        run.step(points[i % npts], False)
    run.fence()
    t0 = time.perf_counter()
    if native_loop:
        ev.eval_each(xs_timed, False)
    else:
        for i in range(steps):
            run.step(points[i % npts], False)
    run.fence()
    elapsed_f = time.perf_counter() - t0
    # ---- beside the headline: the same K evaluations as INDEPENDENT candidates, queued back to back with one wait per
    # list of 16 (cmx_*_eval_many) -- what a caller with a list of trial points gets; a line search cannot use it
    pipelined = None
    if run.world == 1 and run.sh is None and not run.ev_sharded:
        xs = np.vstack([points[i % npts] for i in range(16)])
        ev.eval_many(xs, True)
        reps = max(1, steps // 16)
        run.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            ev.eval_many(xs, True)
        run.fence()
        el_m = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(reps):
            ev.eval_many(xs, False)
        run.fence()
        el_mf = time.perf_counter() - t0
        pipelined = {"evaluations_per_call": 16, "calls": reps, "ms_per_evaluation": el_m / (16 * reps) * 1e3,
                     "value": n_total * 16 * reps / el_m, "unit": "events/s",
                     "cost_only_ms_per_evaluation": el_mf / (16 * reps) * 1e3,
                     "note": "independent evaluations (cmx_*_eval_many): launch chains queued back to back, one host wait per 16; not "
                             "the headline -- the optimiser's evaluations depend on each other"}
    if run.world > 1:
        t = run.torch.tensor([elapsed, elapsed_f], dtype=run.torch.float64, device=run.device)
        run.dist.all_reduce(t, op=run.dist.ReduceOp.MAX)
        elapsed, elapsed_f = float(t[0].item()), float(t[1].item())
    dom_cal_ms = kernel_ms[dom]
    if tim[dom][1] > 0:
        kernel_ms[dom] = tim[dom][0] / tim[dom][1]  # the sample pass right behind the timed region
    dom_agree = abs(kernel_ms[dom] - dom_cal_ms) <= 0.15 * dom_cal_ms  # (two passes of the same launches, either side of the region)
    # front end, round 6: the adjoint image pass rides inside the splat launch (CMX_OPT_FUSED_IMAGE): one kernel class, the bytes of both
    fused = kind == "frontend" and adjoint and stats.get("fused_evals", 0) > 0 and "image" not in kernel_ms

    # ---- per-kernel roofline table
    nnz = getattr(run, "nnz_pixels", 0)
    models = byte_models(kind, order, n_local, npix, nb, P, adjoint, nnz, getattr(run, "image_pixels", None))
    if kind == "backend" and adjoint and "batch" not in kernel_ms and "gather" in models:
        # the per-batch pass ran inside the gather kernel (CMX_OPT_FOLD_BATCH): it reads the 152-byte Jacobian record of every
        # batch there and the 48-byte per-batch partial sums never exist
        a_g, m_g = models["gather"]
        models["gather"] = (a_g + nb * 152, m_g - nb * 48 + nb * 152)
    if fused:
        a_s, m_s = models["splat"]
        a_i, m_i = models["image"]
        models["splat"] = (a_s + a_i, m_s + m_i)
    pmc, pmc_note = load_pmc()
    kernels = []
    for k in sorted(kernel_ms, key=lambda k: -kernel_ms[k]):
        if k == "comm":
            continue
        alg, mand = models.get(k, (0, 0))
        ms = kernel_ms[k]
        row = {"kernel": ("splat+image" if (fused and k == "splat") else k), "ms": ms, "launches_per_eval": per_eval.get(k, 1.0), "alg_bytes": alg,
               "hbm_mandatory_bytes": mand, "pmc_bytes": pmc.get("%s_%s" % (pmc_prefix, k)),
               "frac": mand / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
               "ratio_8d_bytes_to_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
        if row["pmc_bytes"]:
            row["frac_pmc"] = row["pmc_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        kernels.append(row)
    ms_step = elapsed / steps * 1e3
    dom_alg, dom_mand = models.get(dom, (0, 0))
    dms = kernel_ms[dom]
    # the honest per-launch figure: mandatory bytes (<= algorithmic) over the live duration; `frac` <= 1 by construction
    achieved = dom_mand / (dms * 1e-3) / 1e9
    mand_eval = sum(models.get(k, (0, 0))[1] * per_eval.get(k, 1.0) for k in kernel_ms if k != "comm")
    roofline = {"bound": "hbm", "kernel": ("splat+image" if (fused and dom == "splat") else dom), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("%s_%s" % (pmc_prefix, dom)), "traffic_note": pmc_note,
                "model": "hbm-mandatory bytes per launch (coalesced per-event streams + one pass over each plane + one 4-byte "
                         "write per non-zero IWE pixel) / live kernel duration",
                "limited_by": LIMITED_BY.get((kind, dom)),
                "bytes_per_launch": dom_mand, "avg_launch_ms": dms, "avg_launch_ms_calibration": dom_cal_ms,
                "launch_ms_passes_agree": bool(dom_agree), "launch_ms_from": "%d timed launches right behind the timed region "
                "(no launch inside it carries timing events); calibration pass in front of it beside it" % int(tim[dom][1]),
                "events_per_launch": int(n_local),
                "alg_bytes_per_launch_8d": dom_alg, "ratio_8d_bytes_to_peak": dom_alg / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "ratio_8d_bytes_to_peak uses SURVEY 8(d)'s algorithmic bytes (vote read-modify-writes counted as memory "
                        "traffic although they stay in LDS): it is not an HBM fraction and may exceed 1 for large launches"}
    # Price the kernel against what BINDS it.  The back end's per-event kernels issue ~260 VALU instructions per event, most of them
    # fp64 (rotation, atan2 / asin polynomials, fp64 gradient sums): with the SQ counters of this build at hand (profiles/pmc_traffic.json,
    # rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU passes of this command) the headline roofline of a back-end leg is the VALU
    # pipe at the fp64 issue rate; the HBM figure stays beside it (hbm_frac, hbm_achieved).
    valu = pmc.get("%s_%s_valu_insts" % (pmc_prefix, dom))
    if kind == "backend" and valu:
        # Round 6: priced per instruction CLASS.  tools/microbench/valu_rates.hip (profiles/r06_valu_rates.txt): every class issues one
        # wave-instruction per 4 clocks per SIMD on gfx950 -- fp64 FMA / ADD / MUL, fp32, integer, conversions alike -- except the
        # transcendental classes: TRANS_F32 8 clocks, TRANS_F64 16.  So the launch's issue-slot count is SQ_INSTS_VALU plus one extra
        # slot per TRANS_F32 and three per TRANS_F64 (the rocprofv3 --pmc SQ_INSTS_VALU_* passes of this build).
        mix = pmc.get("%s_%s_valu_mix" % (pmc_prefix, dom)) or {}
        slots = valu + 1.0 * mix.get("trans_f32", 0.0) + 3.0 * mix.get("trans_f64", 0.0)
        t_lane = slots * 64 / (dms * 1e-3) / 1e12
        act = pmc.get("%s_%s_valu_active_x4clk" % (pmc_prefix, dom))
        per_ev = 64.0 / max(n_local, 1)
        f64 = sum(mix.get(k, 0.0) for k in ("add_f64", "mul_f64", "fma_f64", "trans_f64"))
        f32 = sum(mix.get(k, 0.0) for k in ("add_f32", "mul_f32", "fma_f32", "trans_f32"))
        if mix:
            roofline["valu_mix_per_event"] = {"fp64": f64 * per_ev, "fp32": f32 * per_ev, "int32": mix.get("int32", 0.0) * per_ev,
                                              "int64": mix.get("int64", 0.0) * per_ev, "cvt": mix.get("cvt", 0.0) * per_ev,
                                              "trans_f64": mix.get("trans_f64", 0.0) * per_ev, "trans_f32": mix.get("trans_f32", 0.0) * per_ev,
                                              "other (moves, selects, compares)": max(0.0, valu - f64 - f32 - mix.get("int32", 0.0) - mix.get("int64", 0.0)
                                                                                      - mix.get("cvt", 0.0)) * per_ev,
                                              "issue_slots": slots * per_ev}
        roofline.update({"bound": "valu_fp64", "achieved": t_lane, "peak": VALU_FP64_PEAK_TLANE, "unit": "Tinstr/s",
                         "frac": t_lane / VALU_FP64_PEAK_TLANE, "hbm_frac": achieved / HBM_PEAK_GBS, "hbm_achieved_gbs": achieved,
                         "valu_insts_per_launch": valu, "valu_insts_per_event": valu * 64 / max(n_local, 1),
                         # the pipe's own busy counter: clocks with a VALU instruction issuing, per SIMD, over the launch's clocks
                         "valu_busy_frac": (act * 4 / 1024 / (dms * 1e-3 * 2.4e9)) if act else None,
                         "model": "VALU issue slots per launch (SQ_INSTS_VALU + 1 x TRANS_F32 + 3 x TRANS_F64: every class issues in 4 clocks per SIMD on "
                                  "gfx950 except the transcendentals, profiles/r06_valu_rates.txt) x 64 lanes / live kernel duration, against one slot per "
                                  "4 clocks per SIMD (39.3 T lane-slots/s); hbm_frac = the HBM-mandatory bytes of the same launch / duration / 8 TB/s"})
    out = {
        "ms_per_step": ms_step,
        "value": n_total * steps / elapsed,
        "cost_only": {"value": n_total * steps / elapsed_f, "unit": "events/s", "ms_per_step": elapsed_f / steps * 1e3,
                      "kernel_ms": kernel_ms_f},
        "kernel_ms": kernel_ms,
        "kernels": kernels,
        "roofline": roofline,
        "whole_evaluation": {
            "ms": ms_step,
            "sum_of_kernels_ms": sum(kernel_ms[k] * per_eval.get(k, 1.0) for k in kernel_ms if k != "comm"),
            "hbm_mandatory_bytes": mand_eval,
            "frac": mand_eval / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "alg_bytes_8d_reference_flow": whole_eval_bytes_8d(kind, order, n_local, npix, P),
            "ratio_8d_bytes_to_peak": whole_eval_bytes_8d(kind, order, n_local, npix, P) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "ratio_8d_bytes_to_peak prices the REFERENCE's data flow (1+P planes scattered, blurred, reduced) at this run's time: the "
                    "adjoint gradient never moves most of those bytes; frac is what this implementation must move"},
        "timed_loop": ("native: one cmx_*_eval_each call issues the K evaluations one after the other (each waited for)" if native_loop
                       else "python: K calls of the sharded evaluator"),
        "timed_region_split": region_split,
        "trajectory_points": npts, "rebins": stats["rebins"], "fallback_frac_last": stats["fallback_frac"],
        "launches_per_evaluation": sum(per_eval.get(k, 1.0) for k in kernel_ms if k != "comm"),
        "fused_image_pass": {"evaluations": stats.get("fused_evals", 0), "repeated_unfused": stats.get("fused_redos", 0)} if kind == "frontend" else None,
        "contrast": c,
    }
    if pipelined:
        out["pipelined"] = pipelined
    if "comm" in kernel_ms:
        n_comm = cal["comm"][1] / ncal
        out["comm"] = {"ms_per_step": kernel_ms["comm"] * n_comm, "collectives_per_step": n_comm,
                       "share_of_step": kernel_ms["comm"] * n_comm / ms_step,
                       "bytes_last_evaluation": stats.get("comm_bytes"), "exchange_set_tiles": stats.get("exchange_tiles"),
                       "exchange_misses": stats.get("exchange_misses"),
                       "note": "RCCL collectives as seen on rank 0's stream in the calibration steps: the exchange plus the wait "
                               "for the slowest rank"}
    out["_last"] = (c, g)
    return out


# ---------------------------------------------------------------------------------------------- workloads
def frontend_workload(args, ctx, per_gpu):
    from cmax_slam_amd import _lib, evaluator, solver, synth
    from cmax_slam_amd.dist import batch_range
    rank, world, dev = ctx["rank"], ctx["world"], ctx["local_rank"]
    p = synth.config2(per_gpu * world)
    beg, end = batch_range(len(p.x), p.batch, rank, world)
    ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=dev)
    ev.set_packet(p.x[beg:end], p.y[beg:end], p.t_ns[beg:end], p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    adjoint = args.mode == "fast"
    (ev.set_fast_path if adjoint else ev.set_reference_path)()
    run = Runner(ev, world, args.comm, ctx["device"], ctx["torch"], ctx["dist"], force=ctx["sharded"])
    if not ctx["sharded"]:
        points = record_trajectory(ev, np.zeros(3), "frontend", solver.FRONTEND)
    else:  # sharded: the trajectory of the whole problem is not available per rank; a line through the solve's range
        points = [np.array([0.6, -0.9, 0.4]) * s for s in (0.0, 0.35, 0.7, 0.9, 1.0)]
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)  # every timed step is a FULL evaluation (the df-after-f reuse is for solves)
    iwe = ev.computeImageOfWarpedEvents(points[-1], blur=False)
    run.nnz_pixels = int(np.count_nonzero(iwe))
    m = measure(run, points, args.steps, args.warmup, "frontend", 0, end - beg, len(p.x), p.W * p.H, 0, 3, adjoint,
                ("frontend_%s" % args.mode) if (world == 1 and per_gpu == 1_000_000) else "none")
    name = "BASELINE config 2: front-end fdf, %d synthetic events/GPU, 640x480 IWE, batch 100, sigma 1, variance" % per_gpu
    return ev, run, p, m, name, "%dx%d" % (p.W, p.H), points


def prior_map_config5(ctx):
    """Config 5's non-zero global map (alpha != 0): IL_old of a neighbouring window, scaled, computed once on rank 0 by the
    evaluator itself and broadcast, so that every rank (and the one-GPU parity evaluator) holds the same bits."""
    from cmax_slam_amd import _lib, evaluator, synth
    torch, dist = ctx["torch"], ctx["dist"]
    Wp, Hp = 4096, 2048
    t = torch.zeros(Hp * Wp, dtype=torch.float32, device=ctx["device"])
    if ctx["rank"] == 0:
        prev = synth.config5(N=300_000, seed=synth.SEED0 + 55)
        be = evaluator.BackendEvaluator(prev.W, prev.H, prev.lut, prev.Wp, prev.Hp, device=ctx["local_rank"])
        be.set_window(prev.x, prev.y, prev.t_ns, prev.order, prev.knots_init, prev.start_ns, prev.dt_ns, prev.num_fixed,
                      prev.t_next_win_beg_ns, prev.batch, prev.sample_rate, prev.sigma, _lib.VARIANCE)
        be.eval(np.zeros(prev.P), False)
        t.copy_(torch.from_numpy(np.ascontiguousarray(be.get_plane(_lib.PLANE_IL_OLD) * 2.5).reshape(-1)))
        be.close()
    if ctx["sharded"] and dist.is_initialized():
        dist.broadcast(t, src=0)
    return t.cpu().numpy().reshape(Hp, Wp)


def _slabs(make, n, per_gpu):
    """The n time slabs of a window, generated on a thread pool (numpy releases the GIL in the heavy parts: 9 s per 5M-event slab
    on one core would be 75 s for config 4's 40M events)."""
    from concurrent.futures import ThreadPoolExecutor
    if n == 1:
        return [make(0, 1, per_gpu)]
    with ThreadPoolExecutor(max_workers=max(1, min(n, usable_cores()))) as ex:
        return list(ex.map(lambda r: make(r, n, per_gpu), range(n)))


def scaling_series(devices, per_gpu, pts, steps, mode, headline=None):
    """ONE invocation, the whole weak-scaling series: config 4's window of k x per_gpu events through a group over
    devices[:k] for k = 1, 2, 4, ... < N (k = 1: a plain context), timed like the headline (K dependent fdf evaluations by one native
    call), with the exchange's share (HIP events around the collectives of a short sampled pass) and parity against the same window on
    ONE context.  The k = N row is the headline's own measurement.  On a one-GPU box `--group-devices 0,0,0,0` runs it dry (members
    share the device: the numbers say nothing about scaling, the code path is the multi-GPU one)."""
    from cmax_slam_amd import _lib, evaluator, synth
    N = len(devices)
    rows = []
    for k in [q for q in (1, 2, 4, 8, 16) if q < N]:
        row = {"n": k, "devices": devices[:k]}
        try:
            w = synth.concat_slabs(_slabs(synth.config4_slab, k, per_gpu))
            args_w = (w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, w.sigma, _lib.VARIANCE, getattr(w, "IG", None))
            ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=devices[0], devices=devices[:k] if k > 1 else None)
            ev.set_window(*args_w)
            (ev.set_fast_path if mode == "fast" else ev.set_reference_path)()
            ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
            xs = np.vstack([pts[i % len(pts)] for i in range(max(steps, 8))])
            t_end = time.perf_counter() + SETTLE_S
            while time.perf_counter() < t_end:
                ev.eval_each(xs[:8], True)
            t0 = time.perf_counter()
            cs, gs = ev.eval_each(xs, True)
            el = time.perf_counter() - t0
            row["fdf_ms"] = el / len(xs) * 1e3
            row["events_per_s_per_gpu"] = len(w.x) * len(xs) / el / k
            if k > 1:
                ev.timing_enable(["comm"])
                ev.timing_get()
                ev.eval_each(xs[:16], True)
                tim = ev.timing_get()
                ev.timing_enable(False)
                row["comm_ms"] = (tim["comm"][0] / 16) if tim.get("comm", (0, 0))[1] else None
                one = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=devices[0])
                one.set_window(*args_w)
                (one.set_fast_path if mode == "fast" else one.set_reference_path)()
                c1, g1 = one.eval(xs[-1], True)
                one.close()
                row["parity_vs_1gpu"] = {"contrast_rel": abs(float(cs[-1]) - c1) / abs(c1),
                                         "grad_rel_inf": float(np.abs(gs[-1] - g1).max() / np.abs(g1).max())}
            else:
                row["comm_ms"] = 0.0
            ev.close()
        except Exception as e:
            row["error"] = repr(e)
        rows.append(row)
    if headline is not None:
        rows.append(headline)
    return rows


def one_gpu_same_workload(device, which, n_slabs, per_gpu, pts, steps, mode):
    """Weak-scaling reference for the N > 1 lines: slab 0 of the SAME window (per_gpu events, the same spline, map and options) on ONE plain
    context -- the per-GPU work of the sharded run without a partner.  value(N) / (N x this) is the efficiency a reader wants;
    the N = 1 headline is another workload (config 2)."""
    from cmax_slam_amd import _lib, evaluator, synth
    w = (synth.config4_slab if which == "config4" else synth.config5_slab)(0, n_slabs, per_gpu)
    ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=device)
    ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE)
    (ev.set_fast_path if mode == "fast" else ev.set_reference_path)()
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    xs = np.vstack([pts[i % len(pts)] for i in range(max(steps, 16))])
    t_end = time.perf_counter() + SETTLE_S
    while time.perf_counter() < t_end:
        ev.eval_each(xs[:16], True)
    t0 = time.perf_counter()
    ev.eval_each(xs, True)
    ms = (time.perf_counter() - t0) * 1e3 / len(xs)
    ev.close()
    return {"workload": "%s slab 0 of %d, %d events, one plain context" % (which, n_slabs, len(w.x)), "fdf_ms": ms,
            "value": len(w.x) / ms * 1e3, "unit": "events/s"}


def backend_workload(args, ctx, which, per_gpu, steps, group_devices=None):
    """which = 'config3' (single window, N=1), 'config4' (time slab per rank), 'config5' (time slab per rank).
    group_devices = [d0, d1, ...]: ONE process, the whole window (all slabs) handed to a group handle (--single-process)."""
    from cmax_slam_amd import _lib, evaluator, solver, synth
    rank, world, dev = ctx["rank"], ctx["world"], ctx["local_rank"]
    IG = None
    n_members = len(group_devices) if group_devices else 1
    if which == "config3":
        w = synth.config3(per_gpu)
    elif which == "config4":
        w = (synth.concat_slabs(_slabs(synth.config4_slab, n_members, per_gpu)) if group_devices
             else synth.config4_slab(rank, world, per_gpu))
    else:
        w = (synth.concat_slabs(_slabs(synth.config5_slab, n_members, per_gpu)) if group_devices
             else synth.config5_slab(rank, world, per_gpu))
        IG = prior_map_config5(ctx)
    w.IG = IG
    ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=dev, devices=group_devices)
    ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE, IG)
    adjoint = args.mode == "fast"
    (ev.set_fast_path if adjoint else ev.set_reference_path)()
    run = Runner(ev, world, args.comm, ctx["device"], ctx["torch"], ctx["dist"], force=ctx["sharded"] and not group_devices)
    if group_devices:
        run.ev_sharded = True            # (no eval_many on a group; its exchange lives inside the handle)
        run.comm_used = "one process, %d member contexts behind one handle (cmx_backend_create_group), %s" % (
            n_members, {1: "RCCL via ncclCommInitAll", 2: "direct peer-to-peer kernels"}.get(ev.group_info()["transport"], "plain context"))
    if not ctx["sharded"]:
        points = record_trajectory(ev, np.zeros(w.P), "backend", solver.BACKEND)
    else:  # every rank must evaluate the same points: a seeded walk of the size of a solve's steps
        rng = np.random.default_rng(77)
        points = [np.zeros(w.P)] + [rng.normal(0, 0.004, w.P) * s for s in (0.3, 0.6, 0.9, 1.0)]
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    if group_devices and n_members > 1:
        ev.eval(points[-1], False)   # (a group has no split-phase interface; member 0 holds the exchanged planes)
    else:
        ev.accumulate(points[-1], False)
    il_old, il_new = ev.get_plane(_lib.PLANE_IL_OLD), ev.get_plane(_lib.PLANE_IL_NEW)
    run.nnz_pixels = int(np.count_nonzero(il_old) + np.count_nonzero(il_new))
    # the image passes skip 64x16 tiles with nothing (votes or global map) within the filter's reach: count what is left
    occ = (il_old != 0) | (il_new != 0)
    if IG is not None:
        occ |= IG != 0
    Hp, Wp = occ.shape
    tiles = occ[:Hp // 16 * 16, :Wp // 64 * 64].reshape(Hp // 16, 16, Wp // 64, 64).any(axis=(1, 3))
    grown = tiles.copy()
    for dy in (-1, 0, 1):      # reach of the 2r = 8 pixel halo: the neighbouring tiles
        for dx in (-1, 0, 1):
            grown |= np.roll(np.roll(tiles, dy, 0), dx, 1)
    run.image_pixels = int(min(grown.sum() * 1024, Wp * Hp))
    n_local = len(w.x) // n_members
    nb = (n_local - 1 + w.batch - 1) // w.batch
    m = measure(run, points, steps, args.warmup, "backend", w.order, n_local, n_local * world * n_members, w.Wp * w.Hp, nb, w.P, adjoint,
                ("backend_%s" % args.mode) if (which == "config3" and per_gpu == 5_000_000) else "none")
    desc = {"config3": "BASELINE config 3: back-end BA fdf, %d synthetic events, cubic 10-knot SO(3) spline (P=21), 1024x1024 pano",
            "config4": "BASELINE config 4: back-end BA sliding window fdf, %d synthetic events/GPU (time slab per rank), cubic 10-knot "
                       "spline (P=21), 1024x1024 pano",
            "config5": "BASELINE config 5: back-end fdf, %d synthetic events/GPU (time slab per rank), 1280x720 sensor, linear K=5 "
                       "(P=15), 4096x2048 map"}[which] % per_gpu
    return ev, run, w, m, desc, "%dx%d" % (w.Wp, w.Hp), points


def parity_vs_one_gpu(ctx, kind, obj, x_first, x, c_sharded, g_sharded, args):
    """The whole problem (all ranks' events, gathered over the process group) evaluated by ONE evaluator without a
    communicator on rank 0; relative differences of contrast and gradient against the sharded evaluation."""
    from cmax_slam_amd import _lib, evaluator
    torch, dist, world, rank, device = ctx["torch"], ctx["dist"], ctx["world"], ctx["rank"], ctx["device"]

    def gather(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device)
        outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, outs, dst=0)
        return np.concatenate([o.cpu().numpy() for o in outs]) if rank == 0 else None
    xs, ys, ts = gather(obj.x, np.int32), gather(obj.y, np.int32), gather(obj.t_ns, np.int64)
    if rank != 0:
        return None
    one = evaluator.BackendEvaluator(obj.W, obj.H, obj.lut, obj.Wp, obj.Hp, device=ctx["local_rank"])
    one.set_window(xs.astype(np.uint16), ys.astype(np.uint16), ts, obj.order, obj.knots_init, obj.start_ns, obj.dt_ns, obj.num_fixed,
                   obj.t_next_win_beg_ns, obj.batch, obj.sample_rate, obj.sigma, _lib.VARIANCE, getattr(obj, "IG", None))
    (one.set_fast_path if args.mode == "fast" else one.set_reference_path)()
    one.eval(x_first, False)  # alpha is fixed by the FIRST evaluation of a window (event_pano_warper.cpp:201-210): same point
    c1, g1 = one.eval(x, True)
    one.close()
    g1, gs = np.asarray(g1), np.asarray(g_sharded)
    return {"contrast_rel": abs(c_sharded - c1) / abs(c1), "grad_rel_inf": float(np.abs(gs - g1).max() / np.abs(g1).max()),
            "tolerance": 1e-5, "events_total": int(len(xs))}


def concurrent_contexts(device_index, p, pts, hbm_bytes_per_eval, counts=(4, 8), seconds=0.4):
    """Several evaluator contexts on ONE GPU, each with its own host thread and stream, each evaluating its own copy of the
    workload (fdf, every evaluation waited for): one context is bound by the latency of its three dependent launches, the GPU
    by its throughput -- what a process serving several event streams (or the reference's front-end and back-end threads,
    src/node.cpp:22) gets.  Not the headline: `value` stays the single-context rate."""
    import threading
    from cmax_slam_amd import _lib, evaluator
    res = []
    for T in counts:
        evs = []
        for _ in range(T):
            ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=device_index)
            ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
            ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
            for k in range(5):
                ev.eval(pts[k % len(pts)], True)
            evs.append(ev)
        n_done = [0] * T
        go = threading.Barrier(T + 1, timeout=60)  # (a worker that died before the start must not hang the bench)
        stop = [0.0]

        def work(k):
            ev, n = evs[k], 0
            go.wait()
            while time.perf_counter() < stop[0]:
                ev.eval(pts[n % len(pts)], True)
                n += 1
            n_done[k] = n
        th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        for t in th:
            t.start()
        stop[0] = time.perf_counter() + seconds + 0.05
        t0 = time.perf_counter()
        go.wait()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        for ev in evs:
            ev.close()
        evals = sum(n_done)
        res.append({"contexts": T, "evaluations": evals, "seconds": el, "value": evals * len(p.x) / el, "unit": "events/s",
                    "ms_per_evaluation_per_context": el / max(min(n_done), 1) * 1e3,
                    "whole_evaluation_frac": evals * hbm_bytes_per_eval / el / 1e9 / HBM_PEAK_GBS})
    return {"runs": res,
            "note": "aggregate of independent fdf evaluations on one GPU (own thread + stream per context, every evaluation waited "
                    "for); whole_evaluation_frac = evaluations/s x HBM-mandatory bytes of one evaluation / 8 TB/s"}


def cmax_solves(n_solves, ev, kind, _lib):
    """CMax iterations per second: full FR-CG solves (the reference's driver loop, host C++) from the reference's own
    start (front end: omega = 0; back end: zero increments on the perturbed knots), image reuse on as in production."""
    ev.set_option(_lib.OPT_REUSE_IMAGE, 1)
    iters = evals = 0
    (ev.setupProblemAndOptimize(np.zeros(3)) if kind == "frontend" else ev.setupProblemAndOptimize())  # one untimed solve (warm-up)
    s0 = ev.stats()
    t0 = time.perf_counter()
    for _ in range(n_solves):
        x, rep = ev.setupProblemAndOptimize(np.zeros(3)) if kind == "frontend" else ev.setupProblemAndOptimize()
        iters += rep["iterations"]
        evals += rep["n_f"] + rep["n_df"]
    el = time.perf_counter() - t0
    s1 = ev.stats()
    # the same solves one at a time on an IDLE device (how a packet's solve starts in the reference's flow: packets arrive at the event
    # rate, not back to back): start -> result on the host.  Back to back, a device-driven solve also pays the evaluation slot its
    # predecessor had queued ahead when it ended (three launches that return at once, ~13 us) -- throughput, not latency.
    lat = lat_iters = 0.0
    try:
        import torch
        for _ in range(n_solves):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, rep1 = ev.setupProblemAndOptimize(np.zeros(3)) if kind == "frontend" else ev.setupProblemAndOptimize()
            lat += time.perf_counter() - t1
            lat_iters += rep1["iterations"]
    except Exception:
        lat = 0.0
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    return {"iters_per_s": iters / el,
            "idle_start": ({"iters_per_s": lat_iters / lat, "ms_per_solve": lat / n_solves * 1e3,
                            "note": "each solve started on an idle device (start -> result on the host); iters_per_s above is back to back"}
                           if lat > 0 else None), "evals_per_s": evals / el, "solves": n_solves, "iters_per_solve": iters / n_solves,
            "ms_per_solve": el / n_solves * 1e3, "final_cost": rep["final_cost"], "solution": [float(v) for v in x[:6]],
            "evals_per_solve": evals / n_solves,
            # gradient evaluations that found the resident image (df after f) / that found their result already in flight
            # (the gradient pass gated on the device behind the cost-only evaluation, cmx_hint_next_df), per solve
            "df_on_resident_image_per_solve": (s1["spec_hits"] - s0["spec_hits"]) / n_solves,
            "df_served_in_flight_per_solve": (s1["gated_hits"] - s0["gated_hits"]) / n_solves,
            # solves whose line search ran ahead of the host on the device (CMX_OPT_CHAIN_SOLVE; front end), slots they queued,
            # solves the host took over after a disagreement with the device's machine
            "device_driven_solves": s1["chain_solves"] - s0["chain_solves"],
            "device_driven_slots_per_solve": (s1["chain_slots"] - s0["chain_slots"]) / n_solves,
            "host_takeovers": s1["chain_takeovers"] - s0["chain_takeovers"]}


def aos_ingest(device, p, reps=5):
    """events from the host's own records (std::vector<dvs_msgs::Event>, 16-byte AoS) to resident-on-device.
    (a) the *_aos entry points: ONE packing pass straight from the records; (b) what a host had to do before them: the conversion
    loop into three SoA vectors (here numpy's strided field copies, the speed of a plain C loop) + the SoA entry point.  The
    reference's own ingest is pushEvent's per-event push_back of the same 16-byte record (ang_vel_estimator.cpp:68-78) and a copy
    of the packet into event_subset_ (:137-147) -- two passes over the records on the CPU and nothing uploaded."""
    from cmax_slam_amd import _lib, evaluator
    n = len(p.x)
    ev = _lib.dvs_events(p.x, p.y, p.t_ns)
    store = evaluator.EventStore(p.W, p.H, 2 * n, device=device)
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=device)

    def best(fn):
        b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            b = min(b, time.perf_counter() - t0)
        return b * 1e3

    def push_aos():
        store.drop_before(store.end)
        store.push_aos(ev)

    def push_soa():
        store.drop_before(store.end)
        store.push(p.x, p.y, p.t_ns)

    conv = {}

    def convert():
        conv["x"], conv["y"] = np.ascontiguousarray(ev["x"]), np.ascontiguousarray(ev["y"])
        conv["t"] = ev["sec"].astype(np.int64) * 1000000000 + ev["nsec"]

    def packet_aos():
        fe.set_packet_aos(ev, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)

    def packet_soa():
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)

    push_aos()
    out = {"events": n, "record": "dvs_msgs::Event (16 bytes: uint16 x, y; uint32 sec, nsec; bool polarity)",
           "store_push_aos_ms": best(push_aos), "store_push_soa_ms": best(push_soa), "aos_to_soa_conversion_ms": best(convert),
           "set_packet_aos_ms": best(packet_aos), "set_packet_soa_ms": best(packet_soa)}
    out["store_push_aos_events_per_s"] = n / (out["store_push_aos_ms"] * 1e-3)
    out["via_conversion_events_per_s"] = n / ((out["aos_to_soa_conversion_ms"] + out["store_push_soa_ms"]) * 1e-3)
    out["note"] = ("push = packing pass on the host pool into pinned staging + upload, complete on return; set_packet = packing pass + "
                   "per-batch times, uploads queued behind it on the context's stream (the call returns; the first evaluation waits for them)")
    store.close()
    fe.close()
    return out


def per_packet_pipeline(device, which, n_packets=8):
    """The per-packet pipeline of the front end (reference: AngVelEstimator's loop, src/frontend/ang_vel_estimator.cpp:68-147:
    a NEW packet every ~10 ms): hand-over, first evaluation (upload + destination-tile sort + streams), FR-CG solve.

      which = "config2"  : 1M-event packets, 640x480, handed over from host arrays (cmx_frontend_set_packet);
      which = "store60k" : 60 000-event packets, 240x180 (ecrot_synth), cut from the device-resident event store
                           (cmx_events_push once, cmx_frontend_set_packet_from per packet).
    sequential = one context, every packet pays its set-up in front of its solve.  pipelined = two contexts: while packet k
    is being solved on one, a helper host thread hands packet k+1 to the other and calls cmx_frontend_prepare (upload, sort,
    streams, chunk table queued on that context's own stream), so the GPU runs the set-up beside the solve."""
    import queue
    import threading
    from cmax_slam_amd import _lib, evaluator, synth
    if which == "config2":
        base = [synth.config2(seed=synth.SEED0 + 2 + 17 * k) for k in range(3)]
        packets = [base[k % 3] for k in range(n_packets)]
        store = None
        W, H, lut = base[0].W, base[0].H, base[0].lut
        cam = (base[0].fx, base[0].fy, base[0].cx, base[0].cy)
    else:
        n_ev = 60_000
        st = synth.event_stream(2.0e6, n_ev * n_packets / 2.0e6 + 1e-4, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=synth.SEED0 + 9)
        W, H, lut, cam = st.W, st.H, st.lut, (st.fx, st.fy, st.cx, st.cy)
        store = evaluator.EventStore(W, H, len(st.x) + 1024, device=device)
        store.push(st.x, st.y, st.t_ns)
        packets = []
        for k in range(n_packets):
            a, b = k * n_ev, (k + 1) * n_ev
            packets.append((a, n_ev, int((st.t_ns[a] + st.t_ns[b - 1]) // 2)))

    def hand_over(ev, pk):
        if store is None:
            ev.set_packet(pk.x, pk.y, pk.t_ns, pk.t_ref_ns, pk.fx, pk.fy, pk.cx, pk.cy, pk.batch, pk.sigma, _lib.VARIANCE)
        else:
            ev.set_packet_from(store, pk[0], pk[1], pk[2], *cam, 100, 1.0, _lib.VARIANCE)

    evs = [evaluator.FrontendEvaluator(W, H, lut, device=device) for _ in range(2)]
    x0 = np.zeros(3)
    # ---- the parts, one context (every figure the mean over the packets after one untimed pass)
    t_set = t_first = t_solve_fresh = t_solve_res = 0.0
    iters = 0
    for rep in range(2):
        t_set = t_first = t_solve_fresh = t_solve_res = 0.0
        iters = 0
        for pk in packets:
            t0 = time.perf_counter()
            hand_over(evs[0], pk)
            t1 = time.perf_counter()
            evs[0].eval(x0, True)
            t2 = time.perf_counter()
            hand_over(evs[0], pk)        # a fresh packet again: the solve below pays the sort inside its first evaluation
            t3 = time.perf_counter()
            _, r = evs[0].setupProblemAndOptimize(x0)
            t4 = time.perf_counter()
            evs[0].setupProblemAndOptimize(x0)   # the same packet resident and sorted: the solve alone
            t5 = time.perf_counter()
            t_set += t1 - t0
            t_first += t2 - t1
            t_solve_fresh += t4 - t3
            t_solve_res += t5 - t4
            iters += r["iterations"]
    n = len(packets)
    set_ms, first_ms, fresh_ms, solve_ms = (t * 1e3 / n for t in (t_set, t_first, t_solve_fresh, t_solve_res))
    seq_ms = set_ms + fresh_ms
    # ---- pipelined: two contexts, a helper thread stages the next packet while this one is being solved
    jobs = queue.Queue()

    def helper():
        while True:
            job = jobs.get()
            if job is None:
                return
            ev, pk, hint, done = job
            hand_over(ev, pk)
            ev.prepare(hint)
            done.set()
    th = threading.Thread(target=helper, daemon=True)
    th.start()
    best = None
    for rep in range(3):
        d0 = threading.Event()
        jobs.put((evs[0], packets[0], x0, d0))
        d0.wait()
        pend = None
        last = x0
        t0 = time.perf_counter()
        for k in range(n):
            if k + 1 < n:
                pend = threading.Event()
                jobs.put((evs[(k + 1) % 2], packets[k + 1], last, pend))
            last, r = evs[k % 2].setupProblemAndOptimize(x0)
            if k + 1 < n:
                pend.wait()
        el = (time.perf_counter() - t0) * 1e3 / n
        best = el if best is None else min(best, el)
    jobs.put(None)
    th.join()
    for e in evs:
        e.close()
    if store is not None:
        store.close()
    n_ev_pk = len(packets[0].x) if store is None else packets[0][1]
    return {"packet": "%d events, %dx%d, %s" % (n_ev_pk, W, H, "host arrays (cmx_frontend_set_packet)" if store is None else
                                               "cut from the device event store (cmx_frontend_set_packet_from)"),
            "packets": n, "set_packet_ms": set_ms, "first_eval_ms": first_ms, "solve_ms": solve_ms,
            "solve_incl_first_sort_ms": fresh_ms, "iters_per_solve": iters / n,
            "sequential": {"ms_per_packet": seq_ms, "packets_per_s": 1e3 / seq_ms, "iters_per_s_incl_setup": iters / n / seq_ms * 1e3,
                           "ratio_to_solve": seq_ms / solve_ms},
            "pipelined": {"ms_per_packet": best, "packets_per_s": 1e3 / best, "iters_per_s_incl_setup": iters / n / best * 1e3,
                          "ratio_to_solve": best / solve_ms,
                          "how": "two contexts; a helper host thread hands packet k+1 over and calls cmx_frontend_prepare (upload, "
                                 "tile sort, streams, chunk table on that context's stream) while packet k is solved"},
            "note": "solves start at omega = 0 like `cmax`; solve_ms = the solve on a resident, already-sorted packet"}


def per_window_pipeline(device, w, n_windows=4, devices=None):
    """The per-window pipeline of the back end (reference: PoseGraphOptimizer's loop, src/backend/pose_graph_optimizer.cpp:131-165,
    244-323: a NEW window every stride): hand-over, first evaluation (upload + pose table + destination-tile sort + streams),
    FR-CG solve -- BASELINE config 3's window, from host arrays (cmx_backend_set_window) and cut from the device event store
    (cmx_backend_set_window_from).  sequential = one context; pipelined = two contexts, a helper host thread hands window k+1
    over and calls cmx_backend_prepare while window k is solved."""
    import queue
    import threading
    from cmax_slam_amd import _lib, evaluator
    out = {}
    args_w = (w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
    # devices = a group's member list: the handles are groups, the store holds a replica per member device and every member cuts
    # its own batch range from it (cmx_events_create_group + cmx_backend_set_window_from on the group's handle)
    store = evaluator.EventStore(w.W, w.H, len(w.x) + 1024, device=device, devices=devices)
    t0 = time.perf_counter()
    store.push(w.x, w.y, w.t_ns)
    out["store_push_ms"] = (time.perf_counter() - t0) * 1e3  # once per stream, not per window: the events arrive as they are sensed
    for source in ("host_arrays", "device_store"):
        def hand_over(ev):
            if source == "host_arrays":
                ev.set_window(w.x, w.y, w.t_ns, *args_w)
            else:
                ev.set_window_from(store, 0, len(w.x), *args_w)
        evs = [evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=device, devices=devices) for _ in range(2)]
        x0 = np.zeros(w.P)
        for rep in range(2):  # (first pass untimed: allocations)
            t_set = t_first = t_fresh = t_res = 0.0
            iters = 0
            for _ in range(n_windows):
                t0 = time.perf_counter()
                hand_over(evs[0])
                t1 = time.perf_counter()
                evs[0].eval(x0, True)
                t2 = time.perf_counter()
                hand_over(evs[0])
                t3 = time.perf_counter()
                _, r = evs[0].setupProblemAndOptimize()
                t4 = time.perf_counter()
                evs[0].setupProblemAndOptimize()
                t5 = time.perf_counter()
                t_set += t1 - t0
                t_first += t2 - t1
                t_fresh += t4 - t3
                t_res += t5 - t4
                iters += r["iterations"]
        n = n_windows
        set_ms, first_ms, fresh_ms, solve_ms = (t * 1e3 / n for t in (t_set, t_first, t_fresh, t_res))
        seq_ms = set_ms + fresh_ms
        jobs = queue.Queue()

        def helper():
            while True:
                job = jobs.get()
                if job is None:
                    return
                ev, done = job
                hand_over(ev)
                ev.prepare(None)
                done.set()
        th = threading.Thread(target=helper, daemon=True)
        th.start()
        best = None
        for rep in range(3):
            d0 = threading.Event()
            jobs.put((evs[0], d0))
            d0.wait()
            t0 = time.perf_counter()
            for k in range(n):
                pend = None
                if k + 1 < n:
                    pend = threading.Event()
                    jobs.put((evs[(k + 1) % 2], pend))
                evs[k % 2].setupProblemAndOptimize()
                if pend is not None:
                    pend.wait()
            el = (time.perf_counter() - t0) * 1e3 / n
            best = el if best is None else min(best, el)
        jobs.put(None)
        th.join()
        for e in evs:
            e.close()
        out[source] = {"set_window_ms": set_ms, "first_eval_ms": first_ms, "solve_ms": solve_ms, "solve_incl_first_sort_ms": fresh_ms,
                       "iters_per_solve": iters / n,
                       "sequential": {"ms_per_window": seq_ms, "ratio_to_solve": seq_ms / solve_ms},
                       "pipelined": {"ms_per_window": best, "ratio_to_solve": best / solve_ms}}
    store.close()
    out["window"] = "%d events, %dx%d pano, order %d, K %d%s" % (len(w.x), w.Wp, w.Hp, w.order, w.K,
                                                                  (", group over devices %s" % devices) if devices else "")
    out["windows"] = n_windows
    out["note"] = ("solve_ms = the solve on a resident, already-sorted window; pipelined = two contexts, a helper host thread runs "
                   "set_window[_from] + cmx_backend_prepare of window k+1 beside the solve of window k")
    return out


def launch_default_shapes(device, solves=5, steps=200):
    """The reference's own back-end operating points (launch/ijrr.launch:27-33, launch/ecrot_handheld.launch:28-34): LINEAR
    spline, dt_knots 0.05 s, window 0.2 s -> K = 5 control poses, P = 15 (first window: no fixed pose) or 12 (one fixed), batches
    of 100, sigma 1; 1024x512 (DAVIS 240x180) or 4096x2048 (1280x720) panorama; 200k and 1M events per window.  Every shape has
    a full-size parity test (tests/test_gpu_launch_defaults.py)."""
    from cmax_slam_amd import _lib, evaluator, synth
    rows = []
    for name, (W, H, f, Wp, Hp, stride) in (("ijrr", (240, 180, 200.0, 1024, 512, 0.1)),
                                            ("ecrot_handheld", (1280, 720, 1000.0, 4096, 2048, 0.2))):
        for n_ev in (200_000, 1_000_000):
            for nf in (1, 0):
                w = synth.backend_window(n_ev, W, H, f, f, (W - 1) / 2.0, (H - 1) / 2.0, Wp, Hp, 2, 5, nf, 0.2, dt_knots=0.05,
                                         seed=synth.SEED0 + 40 + nf, win_stride=stride)
                ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=device)
                ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                              w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
                rng = np.random.default_rng(5)
                pts = np.vstack([rng.normal(0, 0.003, w.P) * s for s in (0.0, 0.3, 0.6, 1.0)] * (steps // 4))
                ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
                ev.eval_each(pts[:40], True)
                t0 = time.perf_counter()
                ev.eval_each(pts, True)
                fdf_ms = (time.perf_counter() - t0) * 1e3 / len(pts)
                t0 = time.perf_counter()
                ev.eval_each(pts, False)
                f_ms = (time.perf_counter() - t0) * 1e3 / len(pts)
                ev.set_option(_lib.OPT_REUSE_IMAGE, 1)
                ev.setupProblemAndOptimize()
                it = evs = 0
                t0 = time.perf_counter()
                for _ in range(solves):
                    _, r = ev.setupProblemAndOptimize()
                    it += r["iterations"]
                    evs += r["n_f"] + r["n_df"]
                el = time.perf_counter() - t0
                ev.close()
                rows.append({"launch": name, "events": n_ev, "pano": "%dx%d" % (Wp, Hp), "P": w.P, "fdf_ms": fdf_ms,
                             "cost_only_ms": f_ms, "events_per_s": n_ev / fdf_ms * 1e3, "solve_ms": el / solves * 1e3,
                             "iters_per_s": it / el, "iters_per_solve": it / solves, "evals_per_solve": evs / solves})
    return {"shapes": rows, "spline": "linear (So3Spline<2>), K = 5, dt_knots 0.05 s, window 0.2 s, batch 100, sigma 1, variance",
            "note": "solves start at zero increments on the perturbed knots like `cmax`; fdf = cmx_backend_eval_each over 4 points"}


def frontend_beside_backend(device, p, w, seconds=0.35):
    """The reference's two threads on one GPU (src/node.cpp:22 + src/cmax_slam.cpp:92): a front-end context and a back-end
    context, own host thread + stream each.  `back_to_back`: both evaluate fdf in a loop (front end at 100 % duty, the worst
    case).  `at_100hz`: the front end solves one packet every 10 ms -- the reference's rate -- beside a back-end solve loop.
    Stream priorities and CU masks were swept (profiles/r04_fe_beside_be.txt): isolation is available through
    cmx_set_cu_mask at the price of a static split; the default measured here is dynamic sharing."""
    import threading
    from cmax_slam_amd import _lib, evaluator
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut, device=device)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=device)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE)
    xf, xb = np.array([0.3, -0.5, 0.2]), np.zeros(w.P)

    def loops(run_fe, run_be, fe_fn, be_fn, secs):
        cnt, lat = [0, 0], [[], []]
        stop = [0.0]
        go = threading.Barrier(int(run_fe) + int(run_be) + 1, timeout=60)

        def work(k, fn):
            go.wait()
            n = 0
            while time.perf_counter() < stop[0]:
                t0 = time.perf_counter()
                fn()
                lat[k].append(time.perf_counter() - t0)
                n += 1
            cnt[k] = n
        th = ([threading.Thread(target=work, args=(0, fe_fn))] if run_fe else []) + \
             ([threading.Thread(target=work, args=(1, be_fn))] if run_be else [])
        for t in th:
            t.start()
        stop[0] = time.perf_counter() + secs + 0.02
        go.wait()
        for t in th:
            t.join()
        return [float(np.mean(v)) * 1e3 if v else None for v in lat]
    res = {}
    # ---- back to back
    for ev, x in ((fe, xf), (be, xb)):
        ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
        for _ in range(10):
            ev.eval(x, True)
    f_solo = loops(True, False, lambda: fe.eval(xf, True), None, seconds / 2)[0]
    b_solo = loops(False, True, None, lambda: be.eval(xb, True), seconds / 2)[1]
    f_bes, b_bes = loops(True, True, lambda: fe.eval(xf, True), lambda: be.eval(xb, True), seconds)
    res["back_to_back"] = {"frontend_fdf_ms": {"solo": f_solo, "beside": f_bes, "ratio": f_bes / f_solo},
                           "backend_fdf_ms": {"solo": b_solo, "beside": b_bes, "ratio": b_bes / b_solo}}
    # ---- the reference's rate: one front-end solve per 10 ms
    for ev in (fe, be):
        ev.set_option(_lib.OPT_REUSE_IMAGE, 1)
    fe.setupProblemAndOptimize(np.zeros(3))
    be.setupProblemAndOptimize()

    def fe_tick():
        t0 = time.perf_counter()
        fe.setupProblemAndOptimize(np.zeros(3))
        fe_tick.lat.append(time.perf_counter() - t0)
        dt = 0.010 - (time.perf_counter() - t0)
        if dt > 0:
            time.sleep(dt)
    fe_tick.lat = []
    loops(True, False, fe_tick, None, 0.25)
    s_solo = float(np.mean(fe_tick.lat)) * 1e3
    bs_solo = loops(False, True, None, lambda: be.setupProblemAndOptimize(), 0.25)[1]
    fe_tick.lat = []
    _, bs_bes = loops(True, True, fe_tick, lambda: be.setupProblemAndOptimize(), 0.5)
    s_bes = float(np.mean(fe_tick.lat)) * 1e3
    res["at_100hz"] = {"frontend_solve_ms": {"solo": s_solo, "beside": s_bes, "ratio": s_bes / s_solo, "solves": len(fe_tick.lat)},
                       "backend_solve_ms": {"solo": bs_solo, "beside": bs_bes, "ratio": bs_bes / bs_solo},
                       "note": "front end: one 1M-event FR-CG solve started every 10 ms (a packet per dt_ang_vel); back end: config-3 "
                               "solves in a loop"}
    res["settings"] = "default streams (no priority, no CU mask): see profiles/r04_fe_beside_be.txt for the sweep"
    # ---- cooperative scheduling (cmx_set_sched_class): the back end holds its NEXT evaluation while the front end's solve is on
    # the device -- the front end waits for at most the one back-end evaluation already running
    try:
        fe.set_sched_class(_lib.SCHED_URGENT)
        be.set_sched_class(_lib.SCHED_BACKGROUND)
        fe_tick.lat = []
        _, bs_co = loops(True, True, fe_tick, lambda: be.setupProblemAndOptimize(), 0.5)
        s_co = float(np.mean(fe_tick.lat)) * 1e3
        p95 = float(np.percentile(fe_tick.lat, 95)) * 1e3
        for ev in (fe, be):
            ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
        f_co, b_co = loops(True, True, lambda: fe.eval(xf, True), lambda: be.eval(xb, True), seconds)
        res["cooperative"] = {"at_100hz": {"frontend_solve_ms": {"solo": s_solo, "beside": s_co, "ratio": s_co / s_solo, "p95": p95,
                                                                   "solves": len(fe_tick.lat)},
                                           "backend_solve_ms": {"solo": bs_solo, "beside": bs_co, "ratio": bs_co / bs_solo}},
                              "back_to_back": {"frontend_fdf_ms": {"solo": f_solo, "beside": f_co, "ratio": f_co / f_solo},
                                               "backend_fdf_ms": {"solo": b_solo, "beside": b_co, "ratio": b_co / b_solo},
                                               "note": "front end at 100 % duty: the back end runs in its 5 ms starvation guard only"},
                              "settings": "front end CMX_SCHED_URGENT, back end CMX_SCHED_BACKGROUND (host-side, evaluation granularity)"}
    except Exception as e:
        res["cooperative"] = {"error": repr(e)}
    fe.close()
    be.close()
    return res


def large_launch(args, ctx, events=16_000_000, steps=40):
    """The SAME front-end kernels on a launch large enough to leave the latency-bound regime (16M events instead of config 2's 1M;
    640x480, everything else as config 2): what fraction of the HBM roofline the per-event kernels reach when dispatch, tail and
    round-trip latencies stop being the launch.  Not the headline (BASELINE's metric is quoted at 1M events): evidence about the
    kernels."""
    import copy
    a = copy.copy(args)
    a.steps, a.warmup = steps, 5
    ev, run, p, m, name, img, pts = frontend_workload(a, ctx, events)
    ev.close()
    return {"workload": name, "events": events, "fdf_ms": m["ms_per_step"], "events_per_s": m["value"],
            "cost_only_ms": m["cost_only"]["ms_per_step"],
            "kernels": [{"kernel": k["kernel"], "ms": k["ms"], "hbm_mandatory_bytes": k["hbm_mandatory_bytes"], "frac": k["frac"]}
                        for k in m["kernels"]],
            "whole_evaluation_frac": m["whole_evaluation"]["frac"],
            "note": "frac = HBM-mandatory bytes / live kernel duration / 8 TB/s, as in `roofline`; the 1M-event headline's fractions are "
                    "set by launch and tail latencies (roofline.limited_by)"}


def group_on_one_device(device, w, steps=200):
    """One-process multi-GPU group (cmx_backend_create_group) exercised on the ONE device this box has: two members sharing the
    GPU, direct transport, against the single context on the same window -- what the group machinery (fan-out to the worker
    thread, pack / unpack of the exchange set, two in-process all-reduces) costs when the collective moves nothing over a link."""
    from cmax_slam_amd import _lib, evaluator
    rng = np.random.default_rng(77)
    pts = np.vstack([rng.normal(0, 0.004, w.P) * s for s in (0.0, 0.3, 0.6, 0.9)] * (steps // 4))
    res = {}
    for name, devs in (("single_context", None), ("group_of_2_on_one_device", [device, device])):
        ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=device, devices=devs)
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, w.sigma, _lib.VARIANCE)
        ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
        cs, gs = ev.eval_each(pts[:40], True)
        t0 = time.perf_counter()
        cs, gs = ev.eval_each(pts, True)
        ms = (time.perf_counter() - t0) * 1e3 / len(pts)
        res[name] = {"fdf_ms": ms, "events_per_s": len(w.x) / ms * 1e3, "contrast": float(cs[-1])}
        # what the spin policy costs (CMX_OPT_SPIN_WAIT; 1 = default: evaluations spin on their ticket, idle threads spin 50 us then sleep;
        # 0 = never spin; 20 = a 20 us budget everywhere): the same loop under each setting
        pol = {}
        for val in (0, 20, 1):
            ev.set_option(_lib.OPT_SPIN_WAIT, val)
            ev.eval_each(pts[:16], True)
            t0 = time.perf_counter()
            ev.eval_each(pts, True)
            pol[{0: "never_spin", 20: "budget_20us", 1: "default"}[val]] = (time.perf_counter() - t0) * 1e3 / len(pts)
        res[name]["fdf_ms_by_spin_policy"] = pol
        if devs:
            st, info = ev.stats(), ev.group_info()
            res[name].update({"comm_bytes_last_evaluation": st["comm_bytes"], "exchange_set_tiles": st["exchange_tiles"],
                              "events_per_member": info["events_per_member"], "transport": "direct (peer kernels + HIP events)",
                              "grad_rel_vs_single": float(np.abs(gs[-1] - res["_g"]).max() / np.abs(res["_g"]).max()),
                              "contrast_rel_vs_single": abs(float(cs[-1]) - res["single_context"]["contrast"]) / abs(res["single_context"]["contrast"])})
        else:
            res["_g"] = gs[-1].copy()
        ev.close()
    res.pop("_g")
    res["overhead_ms"] = res["group_of_2_on_one_device"]["fdf_ms"] - res["single_context"]["fdf_ms"]
    try:  # the window hand-over through the group: host arrays against the replicated device store
        res["per_window"] = per_window_pipeline(device, w, n_windows=3, devices=[device, device])
    except Exception as e:
        res["per_window"] = {"error": repr(e)}
    res["note"] = ("two members on ONE GPU serialise on its compute units: the difference to the single context is the group's machinery "
                   "(collective kernels, event waits, fan-out), not a speed-up; a group of one is a plain context (no overhead)")
    return res


def host_cpu():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return "%s x%d" % (line.split(":", 1)[1].strip(), os.cpu_count() or 1)
    except OSError:
        pass
    return "unknown x%d" % (os.cpu_count() or 1)


def usable_cores():
    """Cores this process may actually run on: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(kind, obj, x0, seconds):
    """The CPU oracle ("port": plain-C restatement of the reference path, 1 thread like the reference) timed on this box's
    host cores on the same workload; bounded to ~`seconds` of CPU work."""
    from oracle import pyoracle as po
    po.build()
    if kind == "frontend":
        ref = po.Frontend(obj.W, obj.H, obj.lut, obj.fx, obj.fy, obj.cx, obj.cy, obj.batch, obj.sigma, po.VARIANCE)
        ref.set_packet(obj.x, obj.y, obj.t_ns, obj.t_ref_ns)
    else:
        ref = po.Backend(obj.W, obj.H, obj.lut, obj.Wp, obj.Hp, obj.order, obj.batch, obj.sample_rate, obj.sigma, po.VARIANCE)
        ref.set_window(obj.x, obj.y, obj.t_ns, obj.knots_init, obj.start_ns, obj.dt_ns, obj.num_fixed, obj.t_next_win_beg_ns)
    ref.eval(x0, True)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        ref.eval(x0, True)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= 2000:
            break
    out = {"value": len(obj.x) * n / el, "unit": "events/s", "cores": 1, "kind": "port",
           "sample": "%d full fdf evaluations of the same %d-event workload (%.1f s), single thread like the reference"
                     % (n, len(obj.x), el),
           "ms_per_step": el / n * 1e3, "host": host_cpu()}
    # beside it, labelled: the same restatement on every host core (thread-private images summed in thread order).
    # NOT the reference -- cmax_slam runs each path on one thread -- and not `cpu_baseline.value`.
    try:
        cores = usable_cores()
        cap = cores if kind == "frontend" else max(1, min(cores, 32))
        best_t, best_ms, t_try = 1, None, cap
        while t_try >= 2:
            ref.eval_allcores(x0, True, t_try)  # warm: thread pool + scratch pages
            t0 = time.perf_counter()
            ref.eval_allcores(x0, True, t_try)
            ms = (time.perf_counter() - t0) * 1e3
            if best_ms is None or ms < best_ms:
                best_t, best_ms = t_try, ms
            elif ms > 2 * best_ms:
                break
            t_try //= 2
        m, t0 = 0, time.perf_counter()
        while True:
            ref.eval_allcores(x0, True, best_t)
            m += 1
            el2 = time.perf_counter() - t0
            if el2 > max(2.0, seconds / 4) or m >= 2000:
                break
        out["allcores"] = {"value": len(obj.x) * m / el2, "unit": "events/s", "cores": best_t, "kind": "port + OpenMP",
                           "note": "not the reference (single-threaded): thread-private images + reduction; thread count = "
                                   "the fastest of usable_cores / 2^k",
                           "ms_per_step": el2 / m * 1e3, "usable_cores": cores}
    except Exception as e:  # a missing libgomp must not cost the headline line
        out["allcores"] = {"error": str(e)}
    return out


def synth_config4_slab():
    from cmax_slam_amd import synth
    return synth.config4_slab(2, 8, 5_000_000)


def summary_of(out):
    """Compact headline figures of every leg (the stdout line carries this; the legs themselves live in bench_detail.json)."""
    def g(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    s = {"fdf_ms": out.get("ms_per_step"), "events_per_s": out.get("value"), "n_gpus": out.get("n_gpus"),
         "cost_only_ms": g(out, "cost_only", "ms_per_step"),
         "roofline_frac": g(out, "roofline", "frac"), "roofline_kernel": g(out, "roofline", "kernel"),
         "whole_evaluation_frac": g(out, "whole_evaluation", "frac"), "cmax_iters_per_s": g(out, "cmax", "iters_per_s"),
         "cmax_iters_per_s_idle_start": g(out, "cmax", "idle_start", "iters_per_s"),
         "cpu_baseline_events_per_s": g(out, "cpu_baseline", "value")}
    if isinstance(out.get("kernel_ms"), dict):
        fz = (out.get("fused_image_pass") or {}).get("evaluations", 0) > 0 and "image" not in out["kernel_ms"]
        s["kernel_us"] = {("splat+image" if (fz and k == "splat") else k): round(v * 1e3, 2) for k, v in out["kernel_ms"].items()}
        s["launches_per_evaluation"] = out.get("launches_per_evaluation")
    if isinstance(out.get("backend"), dict):
        b = out["backend"]
        s["backend"] = {"fdf_ms": b.get("ms_per_step"), "events_per_s": b.get("value"), "roofline": {k: g(b, "roofline", k) for k in
                        ("bound", "kernel", "frac", "hbm_frac", "achieved", "peak", "unit", "valu_insts_per_event", "valu_busy_frac", "avg_launch_ms")
                        if g(b, "roofline", k) is not None},
                        "whole_evaluation_frac": g(b, "whole_evaluation", "frac"),
                        "cmax_iters_per_s": g(b, "cmax", "iters_per_s"),
                        "cpu_baseline_events_per_s": g(b, "cpu_baseline", "value"),
                        "per_window_ratio_to_solve": {"store_seq": g(b, "per_window", "device_store", "sequential", "ratio_to_solve"),
                                                      "store_pipelined": g(b, "per_window", "device_store", "pipelined", "ratio_to_solve"),
                                                      "host_seq": g(b, "per_window", "host_arrays", "sequential", "ratio_to_solve"),
                                                      "host_pipelined": g(b, "per_window", "host_arrays", "pipelined", "ratio_to_solve")}}
        if isinstance(b.get("kernel_ms"), dict):
            s["backend"]["kernel_us"] = {k: round(v * 1e3, 2) for k, v in b["kernel_ms"].items()}
    if "frontend_beside_backend" in out:
        f = out["frontend_beside_backend"]
        s["frontend_beside_backend"] = {"fe_b2b": g(f, "back_to_back", "frontend_fdf_ms", "ratio"),
                                        "be_b2b": g(f, "back_to_back", "backend_fdf_ms", "ratio"),
                                        "fe_solve_100hz": g(f, "at_100hz", "frontend_solve_ms", "ratio"),
                                        "be_solve_100hz": g(f, "at_100hz", "backend_solve_ms", "ratio"),
                                        "coop_fe_solve_100hz": g(f, "cooperative", "at_100hz", "frontend_solve_ms", "ratio"),
                                        "coop_be_solve_100hz": g(f, "cooperative", "at_100hz", "backend_solve_ms", "ratio")}
    if isinstance(out.get("group"), dict):
        if "overhead_ms" in out["group"]:
            s["group_2_members_one_device"] = {"overhead_ms": out["group"].get("overhead_ms"),
                                               "fdf_ms_never_spin_vs_default": [g(out["group"], "group_of_2_on_one_device", "fdf_ms_by_spin_policy", "never_spin"),
                                                                                g(out["group"], "group_of_2_on_one_device", "fdf_ms_by_spin_policy", "default")],
                                               "per_window_ratio_to_solve_store": g(out["group"], "per_window", "device_store", "sequential", "ratio_to_solve"),
                                               "per_window_ratio_to_solve_host": g(out["group"], "per_window", "host_arrays", "sequential", "ratio_to_solve"),
                                               "set_window_ms_store": g(out["group"], "per_window", "device_store", "set_window_ms"),
                                               "set_window_ms_host": g(out["group"], "per_window", "host_arrays", "set_window_ms")}
        else:
            s["group"] = {k: out["group"].get(k) for k in ("members", "devices", "transport", "last_fanout_us", "transport_info")}
    if isinstance(out.get("per_window"), dict):
        s["per_window_ratio_to_solve"] = {"store_seq": g(out, "per_window", "device_store", "sequential", "ratio_to_solve"),
                                          "host_seq": g(out, "per_window", "host_arrays", "sequential", "ratio_to_solve"),
                                          "store_set_window_ms": g(out, "per_window", "device_store", "set_window_ms"),
                                          "host_set_window_ms": g(out, "per_window", "host_arrays", "set_window_ms")}
    if isinstance(out.get("aos_ingest"), dict) and "store_push_aos_ms" in out["aos_ingest"]:
        ai = out["aos_ingest"]
        s["aos_ingest_1M"] = {"push_aos_ms": round(ai["store_push_aos_ms"], 3), "convert_plus_push_soa_ms": round(ai["aos_to_soa_conversion_ms"] + ai["store_push_soa_ms"], 3),
                              "events_per_s": ai["store_push_aos_events_per_s"]}
    if isinstance(out.get("scaling_series"), list):  # the whole series of ONE invocation: n, events/s/GPU, exchange ms, parity
        def _par(r):
            pv = r.get("parity_vs_1gpu") or {}
            return None if not pv else max(pv.get("contrast_rel") or 0.0, pv.get("grad_rel_inf") or 0.0)
        s["scaling_series"] = [{"n": r.get("n"), "events_per_s_per_gpu": r.get("events_per_s_per_gpu"), "comm_ms": r.get("comm_ms"),
                                "parity_vs_1gpu": _par(r), **({"error": r["error"][:60]} if "error" in r else {})} for r in out["scaling_series"]]
    if isinstance(out.get("comm"), dict):
        s["comm"] = {k: out["comm"].get(k) for k in ("nranks_seen", "transport", "ms_per_step", "share_of_step", "collectives_per_step",
                                                     "bytes_last_evaluation")}
    if isinstance(out.get("large_launch"), dict) and "kernels" in out["large_launch"]:
        ll = out["large_launch"]
        s["large_launch_16M_events"] = {"events_per_s": ll.get("events_per_s"), "whole_evaluation_frac": ll.get("whole_evaluation_frac"),
                                        "kernel_fracs": {k["kernel"]: k["frac"] for k in ll["kernels"]}}
    if isinstance(out.get("concurrent_contexts"), dict) and out["concurrent_contexts"].get("runs"):
        r = out["concurrent_contexts"]["runs"][-1]
        s["concurrent_contexts"] = {"contexts": r["contexts"], "events_per_s": r["value"], "whole_evaluation_frac": r["whole_evaluation_frac"]}
    if "parity_vs_1gpu" in out:
        s["parity_vs_1gpu"] = {k: g(out, "parity_vs_1gpu", k) for k in ("contrast_rel", "grad_rel_inf", "tolerance", "error")
                               if g(out, "parity_vs_1gpu", k) is not None}
    if isinstance(out.get("config5"), dict):
        c5 = out["config5"]
        s["config5"] = {"fdf_ms": c5.get("ms_per_step"), "events_per_s": c5.get("value"),
                        "parity_grad_rel_inf": g(c5, "parity_vs_1gpu", "grad_rel_inf"), "error": c5.get("error")}
    if isinstance(out.get("one_gpu_same_workload"), dict):
        s["one_gpu_same_workload_events_per_s"] = out["one_gpu_same_workload"].get("value")
    if isinstance(out.get("process_per_gpu"), dict):
        c = out["process_per_gpu"]
        s["process_per_gpu"] = ({"error": c["error"][:200]} if "error" in c and "value" not in c else
                                {"n_gpus": c.get("n_gpus"), "fdf_ms": c.get("ms_per_step"), "events_per_s": c.get("value"),
                                 "nranks_seen": g(c, "summary", "comm", "nranks_seen"),
                                 "parity_grad_rel_inf": g(c, "summary", "parity_vs_1gpu", "grad_rel_inf")})
    ps = parity_sweep_record()
    if ps:
        s["parity_sweep"] = ps
    return s


def parity_sweep_record():
    """profiles/sweep300.json: how many of the back-end sweep's evaluations pass the strict 1e-5 gate against the oracle and how many need
    the all-fp64 arbiter (written by tests/test_gpu_sweep300.py on the GPU box -- the bench itself never touches the oracle outside
    cpu_baseline); reported with whether it was taken on this build."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "sweep300.json")))
        return {"evaluations": rec["evaluations"], "via_arbiter": rec["via_arbiter"], "max_rel_vs_oracle": rec.get("max_rel_vs_oracle"),
                "this_build": rec.get("src_sha256") == csrc_hash()}
    except Exception:
        return None


LINE_BUDGET = 4096  # bytes of the ONE stdout line (round 4's 21 KB line was not parsed by the driver)
DTYPE = "f64"  # the arithmetic the path computes in: fp64 warp, moments and gradient sums (planes are f32, votes 64-bit fixed point)
DTYPE_NOTE = ("f64 warp; votes in 64-bit 2^-30 fixed point (LDS) -> f32 planes; f32 blur (f64-accumulated adjoint operator); f64 moments "
              "and gradient sums")


def plan_launch(gpus, world_env, n_devices, single_process=False, group_devices=None):
    """Which form of the bench a command line asks for -- pure logic, tested on the CPU (tests/test_bench_helpers.py).

      world_env > 1            : launched by torch.distributed.run -> one process per GPU ("ranks"); n_gpus = the world size
      --group-devices a,b,...  : one process, a group handle over exactly those members ("group")
      --gpus N > 1, no launcher: one process, a group over devices 0..N-1 ("group") and the process-per-GPU form self-spawned
                                 as a nested leg (spawn = N); with fewer visible devices than N: whatever is there (one device =
                                 config 4's slab on a plain context), n_gpus = what ran, `error` says so, nothing spawned
      otherwise                : the single-GPU line ("single")."""
    if world_env > 1:
        err = None if gpus in (1, world_env) else "--gpus %d ignored: launched with WORLD_SIZE=%d" % (gpus, world_env)
        return {"form": "ranks", "devices": None, "n_gpus": world_env, "spawn": 0, "error": err}
    if group_devices:
        devs = [int(d) for d in group_devices]
        bad = [d for d in devs if d < 0 or d >= max(n_devices, 0)]
        if bad:
            return {"form": "none", "devices": [], "n_gpus": 0, "spawn": 0,
                    "error": "--group-devices names device(s) %s; %d visible" % (bad, n_devices)}
        return {"form": "group", "devices": devs, "n_gpus": len(set(devs)), "spawn": 0, "error": None}
    if n_devices <= 0:
        return {"form": "none", "devices": [], "n_gpus": 0, "spawn": 0, "error": "no HIP device visible"}
    if gpus > 1 or single_process:
        if n_devices >= gpus:
            return {"form": "group" if gpus > 1 else "single", "devices": list(range(gpus)) if gpus > 1 else None, "n_gpus": gpus,
                    "spawn": 0 if single_process or gpus == 1 else gpus, "error": None}
        return {"form": "group", "devices": list(range(n_devices)), "n_gpus": n_devices, "spawn": 0,
                "error": "--gpus %d requested, %d device(s) visible: ran on %d" % (gpus, n_devices, n_devices)}
    return {"form": "single", "devices": None, "n_gpus": 1, "spawn": 0, "error": None}


def leg_errors(out, path="", acc=None):
    """Every nested {"error": ...} a leg's try/except left behind, as "path: text" strings: they go into the stdout line."""
    acc = [] if acc is None else acc
    if isinstance(out, dict):
        for k, v in out.items():
            if k == "error" and isinstance(v, str) and path:
                acc.append("%s: %s" % (path, v[:120]))
            elif isinstance(v, (dict, list)):
                leg_errors(v, (path + "." if path else "") + str(k), acc)
    elif isinstance(out, list):
        for i, v in enumerate(out):
            leg_errors(v, "%s[%d]" % (path, i), acc)
    return acc


def _r(v, sig=5):
    """Floats to `sig` significant digits (the line is read by a machine; twelve digits of a timer are noise)."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "hbm_frac",
                 "valu_insts_per_event", "valu_busy_frac")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample", "ms_per_step", "host")
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")


def compact_line(out, detail_name="bench_detail.json"):
    """The ONE stdout line: the contract's keys + roofline + cpu_baseline + summary, <= LINE_BUDGET bytes whatever the legs
    produced (optional parts are dropped in a fixed order until it fits; the contract's keys never are)."""
    d = {k: out.get(k) for k in CONTRACT_KEYS}
    cfg = dict(out.get("config") or {})
    for k, n in (("workload", 200), ("parallelism", 160), ("parameters", 80), ("mode", 60)):
        if isinstance(cfg.get(k), str):
            cfg[k] = cfg[k][:n]
    d["config"] = cfg
    if "error" in out:
        d["error"] = str(out["error"])[:300]
    if isinstance(out.get("roofline"), dict):
        d["roofline"] = {k: out["roofline"].get(k) for k in ROOFLINE_KEYS if k in out["roofline"]}
    if isinstance(out.get("cpu_baseline"), dict):
        d["cpu_baseline"] = {k: out["cpu_baseline"].get(k) for k in CPU_BASELINE_KEYS if k in out["cpu_baseline"]}
    d["summary"] = summary_of(out)
    errs = leg_errors({k: v for k, v in out.items() if k != "error"})
    if errs:
        d["leg_errors"] = errs[:6]
    d["detail"] = detail_name
    d = _r(d)
    # never over budget: optional parts go first, in this order
    drop = (("summary", "large_launch_16M_events"), ("summary", "frontend_beside_backend"), ("summary", "process_per_gpu"),
            ("summary", "backend"), ("leg_errors",), ("cpu_baseline", "host"), ("cpu_baseline", "sample"), ("config", "parameters"),
            ("config", "parallelism"), ("summary",))
    for path in drop:
        if len(json.dumps(d, separators=(",", ":"))) <= LINE_BUDGET:
            break
        t = d
        for k in path[:-1]:
            t = t.get(k) if isinstance(t, dict) else None
        if isinstance(t, dict):
            t.pop(path[-1], None)
    return json.dumps(d, separators=(",", ":"))


def spawn_process_per_gpu(n, args, timeout_s=900):
    """The process-per-GPU form of the same command as a child job: `python -m torch.distributed.run --nproc-per-node n bench.py
    --gpus n ...`; returns the child's stdout line (parsed) or {"error": ...}.  Runs after this process has released its contexts."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--mode", args.mode, "--comm", args.comm, "--detail-out", os.path.splitext(args.detail_out)[0] + "_ranks.json"]
    if args.events:
        cmd += ["--events", str(args.events)]
    if args.no_config5:
        cmd += ["--no-config5"]
    if args.no_parity:
        cmd += ["--no-parity"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "process-per-GPU child timed out after %d s" % timeout_s}
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip().startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "process-per-GPU child rc %d: %s" % (p.returncode, p.stderr.strip()[-300:])}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": "process-per-GPU child line did not parse: %s" % e}


def line(m, world, args, name, n_total, img, comm_used, mode_desc):
    out = {
        "metric": METRIC, "value": m["value"], "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE, "dtype_note": DTYPE_NOTE, "data": "synthetic",
        "config": {"workload": name, "events_total": int(n_total), "image": img, "evaluation": "cost+gradient (fdf)", "mode": mode_desc,
                   "parameters": "cycled through %d points of a recorded FR-CG solve" % m["trajectory_points"],
                   "parallelism": ("events sharded by contiguous batch range (time slab) x%d, all-reduce of the partial planes + partial "
                                   "gradient sums; %s" % (world, comm_used)) if world > 1 else "single GPU"},
        "per_gpu_value": m["value"] / world,
    }
    for k in ("cost_only", "pipelined", "kernel_ms", "kernels", "roofline", "whole_evaluation", "comm", "rebins", "fallback_frac_last", "contrast",
              "timed_loop", "timed_region_split", "launches_per_evaluation", "fused_image_pass"):
        if k in m:
            out[k] = m[k]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--steps-backend", type=int, default=None, help="timed steps of the nested back-end leg at N=1 (default: --steps)")
    ap.add_argument("--workload", default="auto", choices=["auto", "frontend", "backend"],
                    help="auto: N=1 -> front end (config 2) with the back end (config 3) nested; N>1 -> back end config 4 with "
                         "config 5 nested.  frontend / backend force one family (N=1 backend = config 3 as the headline)")
    ap.add_argument("--events", type=int, default=None, help="events per GPU (default: the config's own)")
    ap.add_argument("--mode", default="fast", choices=["fast", "faithful"],
                    help="fast = adjoint gradient + LDS-privatised splat (production path); faithful = derivative planes + "
                         "one global atomic per vote (the reference's data flow)")
    ap.add_argument("--comm", default="native", choices=["native", "torch"])
    ap.add_argument("--solves", type=int, default=20, help="FR-CG solves timed for the CMax iters/s figure (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-packet", action="store_true", help="N=1: skip the per-packet pipeline measurement")
    ap.add_argument("--no-backend", action="store_true", help="N=1: skip the nested config-3 leg")
    ap.add_argument("--no-config5", action="store_true", help="N>1: skip the nested config-5 leg")
    ap.add_argument("--no-parity", action="store_true", help="N>1: skip the parity check against one GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process: a group handle (cmx_backend_create_group) over devices 0..N-1, one host thread, one "
                         "optimiser -- the form the reference's single-process host can use; launch WITHOUT torch.distributed.run")
    ap.add_argument("--group-devices", default=None,
                    help="--single-process with an explicit member list, e.g. 0,0 = two members sharing device 0 (how a one-GPU box "
                         "exercises the group path end to end)")
    ap.add_argument("--no-extras", action="store_true", help="N=1: skip per_window / launch_defaults / frontend_beside_backend / group")
    ap.add_argument("--no-spawn", action="store_true", help="plain --gpus N: skip the self-spawned process-per-GPU leg")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where everything that is not on the stdout line goes (JSON)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="dry run of the N>1 code path (time slabs, communicator, parity gather) with whatever world size is launched")
    args = ap.parse_args()
    if args.steps_backend is None:
        args.steps_backend = args.steps

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from cmax_slam_amd import _lib

    n_devices = int(_lib.lib().cmx_device_count())
    plan = plan_launch(args.gpus, world, n_devices, args.single_process,
                       [int(d) for d in args.group_devices.split(",")] if args.group_devices else None)
    if plan["form"] == "none":  # nothing can run: still ONE parseable line and exit status 0 (the driver records the reason)
        out = {"metric": METRIC, "value": 0.0, "unit": "events/s", "n_gpus": 0, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
               "config": {"workload": "none"}, "error": plan["error"]}
        if rank == 0:
            emit(out, args)
        return
    group_devices = plan["devices"] if plan["form"] == "group" else None
    if group_devices is not None and len(group_devices) == 1 and args.workload == "auto":
        args.workload = "backend"  # (--gpus N on a one-GPU box: config 4's slab on the one device, not the N = 1 headline)
    args.gpus = plan["n_gpus"]

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded or group_devices is not None
    if sharded and group_devices is None:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    ctx = dict(rank=rank, world=world, local_rank=local_rank, device=device, torch=torch, dist=dist, sharded=sharded)
    mode_desc = args.mode + (" (adjoint gradient, LDS-privatised splat)" if args.mode == "fast" else " (derivative planes, global atomics)")
    family = args.workload
    if family == "auto":
        family = "backend" if sharded else "frontend"

    out = None
    if family == "frontend":
        ev, run, p, m, name, img, pts = frontend_workload(args, ctx, args.events or 1_000_000)
        if rank == 0:
            out = line(m, world, args, name, len(p.x), img, run.comm_used, mode_desc)
            if world == 1 and args.solves > 0:
                out["cmax"] = cmax_solves(args.solves, ev, "frontend", _lib)
            if world == 1 and not args.no_per_packet:
                try:
                    out["per_packet"] = {"config2": per_packet_pipeline(local_rank, "config2", 6),
                                         "store60k": per_packet_pipeline(local_rank, "store60k", 16)}
                except Exception as e:  # must not cost the headline line
                    out["per_packet"] = {"error": repr(e)}
            if world == 1 and not args.no_per_packet:
                try:
                    out["aos_ingest"] = aos_ingest(local_rank, p)
                except Exception as e:
                    out["aos_ingest"] = {"error": repr(e)}
            if world == 1 and not args.no_per_packet:
                try:
                    out["concurrent_contexts"] = concurrent_contexts(local_rank, p, pts, out["whole_evaluation"]["hbm_mandatory_bytes"])
                except Exception as e:
                    out["concurrent_contexts"] = {"error": repr(e)}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline("frontend", p, pts[len(pts) // 2], args.cpu_seconds)
        ev.close()
        if world == 1 and args.workload == "auto" and not args.no_backend:
            ev, run, w, mb, name_b, img_b, pts_b = backend_workload(args, ctx, "config3", 5_000_000, args.steps_backend)
            be = {"config": {"workload": name_b, "events_total": len(w.x), "image": img_b, "mode": mode_desc},
                  "value": mb["value"], "unit": "events/s", "ms_per_step": mb["ms_per_step"], "steps": args.steps_backend}
            for k in ("cost_only", "pipelined", "kernel_ms", "kernels", "roofline", "whole_evaluation", "rebins", "fallback_frac_last", "timed_loop"):
                if k in mb:
                    be[k] = mb[k]
            if args.solves > 0:
                be["cmax"] = cmax_solves(max(1, args.solves // 2), ev, "backend", _lib)
            if not args.no_cpu_baseline:
                be["cpu_baseline"] = cpu_baseline("backend", w, pts_b[len(pts_b) // 2], max(3.0, args.cpu_seconds / 3))
            out["backend"] = be
            ev.close()
            if not args.no_extras:
                for key, fn in (("per_window", lambda: per_window_pipeline(local_rank, w)),
                                ("launch_defaults", lambda: launch_default_shapes(local_rank))):
                    try:
                        be[key] = fn()
                    except Exception as e:  # must not cost the headline line
                        be[key] = {"error": repr(e)}
                for key, fn in (("frontend_beside_backend", lambda: frontend_beside_backend(local_rank, p, w)),
                                ("group", lambda: group_on_one_device(local_rank, synth_config4_slab())),
                                ("large_launch", lambda: large_launch(args, ctx))):
                    try:
                        out[key] = fn()
                    except Exception as e:
                        out[key] = {"error": repr(e)}
    else:
        which = "config4" if sharded else "config3"
        per_gpu = args.events or 5_000_000
        ev, run, w, m, name, img, pts = backend_workload(args, ctx, which, per_gpu, args.steps, group_devices)
        c, g = m.pop("_last")
        par = None
        if group_devices and not args.no_parity:  # the whole window on ONE context of this process: what the group must equal
            try:
                from cmax_slam_amd import evaluator as _e
                one = _e.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, device=0)
                one.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                               w.sample_rate, w.sigma, _lib.VARIANCE, w.IG)
                one.eval(pts[0], False)
                c1, g1 = one.eval(pts[(args.steps - 1) % len(pts)], True)
                one.close()
                par = {"contrast_rel": abs(c - c1) / abs(c1), "grad_rel_inf": float(np.abs(np.asarray(g) - g1).max() / np.abs(g1).max()),
                       "tolerance": 1e-5, "events_total": int(len(w.x))}
            except Exception as e:
                par = {"error": str(e)}
        elif sharded and not args.no_parity:
            try:
                par = parity_vs_one_gpu(ctx, "backend", w, pts[0], pts[(args.steps - 1) % len(pts)], c, g, args)
            except Exception as e:
                par = {"error": str(e)}
        if rank == 0:
            out = line(m, len(set(group_devices)) if group_devices else world, args, name, len(w.x) * world, img, run.comm_used, mode_desc)
            if par is not None:
                out["parity_vs_1gpu"] = par
            if sharded and "comm" in out:
                try:  # what the communicator itself says it spans (not what the launcher believes)
                    ci = ev.comm_info()
                    out["comm"]["nranks_seen"] = ci["nranks"] if ci["transport"] != "none" else 1
                    out["comm"]["transport"] = ci["transport"]
                except Exception as e:
                    out["comm"]["nranks_seen"] = None
                    out["comm"]["error"] = repr(e)
            if group_devices:
                out["group"] = ev.group_info()
                try:  # how the transport was picked (CMX_GROUP_AUTO: measured at creation) and both candidates' timings
                    out["group"]["transport_info"] = ev.group_transport_info()
                except Exception as e:
                    out["group"]["transport_info"] = {"error": repr(e)}
                out["config"]["form"] = "one process, group handle over devices %s" % group_devices
                if len(group_devices) > 1 and not args.no_extras:
                    try:
                        out["per_window"] = per_window_pipeline(local_rank, w, n_windows=3, devices=group_devices)
                    except Exception as e:
                        out["per_window"] = {"error": repr(e)}
                if args.solves > 0:
                    out["cmax"] = cmax_solves(max(1, args.solves // 4), ev, "backend", _lib)
            if not sharded and args.solves > 0:
                out["cmax"] = cmax_solves(args.solves, ev, "backend", _lib)
            if not sharded and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline("backend", w, pts[len(pts) // 2], args.cpu_seconds)
        ev.close()
        if sharded and not args.no_config5 and not group_devices:
            try:
                ev, run, w5, m5, name5, img5, pts5 = backend_workload(args, ctx, "config5", (args.events or 20_000_000 // 8), args.steps)
                c5, g5 = m5.pop("_last")
                par5 = None
                if not args.no_parity:
                    try:
                        par5 = parity_vs_one_gpu(ctx, "backend", w5, pts5[0], pts5[(args.steps - 1) % len(pts5)], c5, g5, args)
                    except Exception as e:
                        par5 = {"error": str(e)}
                if rank == 0:
                    o5 = line(m5, world, args, name5, len(w5.x) * world, img5, run.comm_used, mode_desc)
                    if par5 is not None:
                        o5["parity_vs_1gpu"] = par5
                    out["config5"] = {k: v for k, v in o5.items() if k not in ("metric", "higher_is_better", "vs_baseline", "dtype", "data")}
                ev.close()
            except Exception as e:
                if rank == 0:
                    out["config5"] = {"error": str(e)}
    if sharded and group_devices is None:
        dist.destroy_process_group()  # (every rank arrives here together; what follows is rank 0's own work)
    if family != "frontend" and group_devices and len(group_devices) > 1 and rank == 0 and not args.no_extras and which == "config4":
        try:
            head = {"n": len(group_devices), "devices": list(group_devices), "fdf_ms": out.get("ms_per_step"),
                    "events_per_s_per_gpu": (out.get("value") or 0.0) / len(group_devices),
                    "comm_ms": (out.get("comm") or {}).get("ms_per_step"), "parity_vs_1gpu": out.get("parity_vs_1gpu")}
            out["scaling_series"] = scaling_series(list(group_devices), per_gpu, pts, args.steps, args.mode, head)
        except Exception as e:
            out["scaling_series"] = {"error": repr(e)}
    if family != "frontend" and sharded and rank == 0 and not args.no_extras:
        n_slabs = len(group_devices) if group_devices else world
        try:
            out["one_gpu_same_workload"] = one_gpu_same_workload(local_rank, which, n_slabs, per_gpu, pts, args.steps, args.mode)
        except Exception as e:
            out["one_gpu_same_workload"] = {"error": repr(e)}
    if rank == 0:
        out.pop("_last", None)
        if plan["error"]:
            out["error"] = plan["error"]
        if plan["spawn"] > 1 and not args.no_spawn:
            # the process-per-GPU form of the same command (one rank per device over RCCL), as a child job now that this process holds no context
            child = spawn_process_per_gpu(plan["spawn"], args)
            out["process_per_gpu"] = child
        emit(out, args)


def emit(out, args):
    """Detail -> bench_detail.json (+ one stderr line); the compact line -> the real stdout, and nothing else there."""
    detail_name = os.path.basename(args.detail_out)
    out["summary"] = summary_of(out)
    try:
        with open(args.detail_out, "w") as f:
            json.dump(out, f)
    except OSError as e:
        detail_name = "stderr (%s)" % e
    text = compact_line(out, detail_name)
    try:  # whatever native libraries still hold in their C stdio buffers goes where stdout currently points: stderr
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    sys.stderr.write("bench detail: " + json.dumps(out) + "\n")
    sys.stderr.flush()
    if _STDOUT_FD is not None:
        os.dup2(_STDOUT_FD, 1)  # the real stdout back: it carries the ONE JSON line and nothing else
    print(text, flush=True)


_STDOUT_FD = None

if __name__ == "__main__":
    # RCCL prints its version banner (six lines) to stdout through C stdio when the first communicator comes up: everything any
    # library prints while the bench runs is sent to stderr, and stdout is handed back for the one JSON line
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    main()
