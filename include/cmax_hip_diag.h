/* cmax_hip_diag.h -- diagnostic option keys of cmx_set_option (include/cmax_hip.h).
 *
 * NOT part of the supported host surface: these keys switch between internal forms of one evaluation that the library selects by
 * itself from the configuration (image size, blur radius, batch size, deterministic mode, communicator attached), always defaulting
 * to the fastest.  They exist so that the parity tests and same-box A/B measurements (tools/ab_eval.py, bench.py) can reach every
 * form on any input; a host integrating the library never sets them, and they may change or disappear between
 * releases without an ABI revision.  Results never depend on them beyond floating-point summation order. */
#ifndef CMAX_HIP_DIAG_H
#define CMAX_HIP_DIAG_H
#include "cmax_hip.h"

enum {
  CMX_OPT_REUSE_IMAGE = 3, /* 1 (default): with CMX_GRAD_ADJOINT, a gradient evaluation at exactly the parameters of
                             the previous evaluation reuses the resident image (GSL's conjugate_fr calls f and then
                             df at every accepted point; the reference recomputes everything, :58-70).  A cost-only
                             evaluation then also runs the adjoint image pass (Jt) instead of the moments-only pass, so the
                             df that follows launches its gather at once: +3..4 us per f, -12 us per df */
  CMX_OPT_TAIL_FINALIZE = 6, /* 1 (default): the last kernel of an evaluation (cost-only: the blur + moments pass; adjoint gradient:
                             the gather pass / the back end's per-batch pass) runs the finalize step -- contrast, gradient,
                             result hand-off -- in its last-arriving workgroup (write-through partial sums, tickets sharded
                             by XCD, sc1 loads) instead of a separate one-workgroup launch behind a kernel boundary.  The
                             gradient sums of the workgroups reach it through 8 rows of accumulators (device-scope fp64
                             atomic adds; their order varies run to run, like the vote image's in this mode); with
                             CMX_OPT_DETERMINISTIC the front end uses a [column][workgroup] table instead and back-end
                             gradient evaluations keep the separate launch (the 42-column table made the tail slower).
                             2: tail with the table form everywhere (back-end gradient included).
                             3: as 1, with a POLLING tail on the front-end gather (workgroup 0 polls sharded fire-and-forget
                                arrival counts and finalizes; round 6 A/B: 14.2 vs 14.1 us, no gain -- and 21.6 us with ONE
                                counter: ~1000 atomics on one memory-side address serialise at ~12 ns each).
                             0: separate finalize launch (the round-1 flow) */
  /* 7: retired (round 2's opt-in fused gradient pass: measured slower, removed; profiles/r02_pmc_fe_fused_gather.txt) */
  CMX_OPT_COMPOSITE_IMAGE = 8, /* 1 (default): the image pass of the adjoint gradient applies G^T G as one banded operator per
                             axis, its 4r+1-term sums accumulated in fp64: three barrier-separated phases per tile instead
                             of five (radius 4 = the reference's blur_sigma 1 has a register-resident form), and a gradient
                             that stays within 1e-5 of the exact-arithmetic value of the reference's formula where long fp32
                             sums do not.  B and the contrast are unchanged to the bit.
                             0: the four-pass fp32 form */
  CMX_OPT_FOLD_BATCH = 9, /* 1 (default; back end, adjoint gradient, batches of a multiple of four events, tail finalize on, not
                             deterministic): the per-batch pass of the gradient (batch Jacobian applied to the batch's sums) runs
                             inside the per-event gather kernel, which then also finalizes -- one launch instead of three.
                             0: separate per-batch kernel */
  CMX_OPT_CHAIN_SOLVE = 11, /* 1 (default; front end, production path, no communicator): cmx_frontend_solve runs the FR-CG line
                               search AHEAD of the host -- the optimiser's state machine lives in device memory, the finalize step
                               of every evaluation advances it and writes the next evaluation point where the next evaluation's
                               kernels (queued one slot ahead) read it; the host replays the machine on the reported costs /
                               gradients and takes over on any disagreement, so the result is that of the host-driven solve.
                               0: host-driven solve (one round trip to the host per evaluation); 2 / 3: test hooks -- the host takes
                               over after three points / between a cost and its gradient, as it would after a disagreement; 4: the first form of
                               the slots (a finalize behind the image pass, a flag-gated gradient pass) also where the self-gating
                               form applies (A/B) */
  CMX_OPT_GATED_DF = 10,  /* 1 (default): act on cmx_hint_next_df (below).  0: ignore the hints */
  CMX_OPT_FUSED_IMAGE = 12 /* 1 (default; front end, production path, blur_sigma 1, no communicator, not deterministic): the adjoint
                               image pass runs INSIDE the splat launch, tile by tile, as the chunk workgroups that can vote into a
                               tile's neighbourhood complete (tile-dataflow fusion): a gradient evaluation
                               is two launches instead of three, and so is a slot of the device-driven solve.  An evaluation with
                               votes beyond the reach the tiles' arrival counts cover is repeated after a fresh sort.
                               2: ONE launch per gradient evaluation -- the gradient gather (its workgroups warp their events while
                               the splat is still running and wait for the tiles their votes touch) and the finalize step ride in
                               the same launch as well.  Built, parity-tested and measured SLOWER than 1 (0.048 vs 0.0365 ms per
                               1M-event evaluation: the three roles compete for the same CUs and the gather's workgroups do not
                               all fit beside the strips; profiles/r06_fused_ab.txt) -- kept as an A/B switch.
                               3: ONE launch of the chunk workgroups ALONE (self-service, cmx_selfserve.hpp): each runs the image
                               pass of the tile(s) it owns, then gathers the gradient sums of its own events, the last arriver
                               finalizes.  Used only once the chunk table's exact length is known and fits the device (all workgroups
                               resident at once; 512 on an MI355X), evaluations it does not cover take form 1; a context whose launches
                               run into the bounded waits three times (it shares the GPU) stops using it.  Parity-tested; SLOWER than 1
                               (0.054 vs 0.0353 ms per 1M-event evaluation): reading the tiles' Jt inside the launch that wrote it needs
                               an agent-scope acquire per workgroup (an L2 invalidate) -- with plain cached loads the launch takes 30 us,
                               a tie with 1, and is wrong about once in 2000 evaluations when contexts share the GPU
                               (profiles/r06_selfserve.txt) -- kept as an A/B switch.
                               0: splat, image pass and gather as three launches */
};

#if defined(__cplusplus)
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Static scheduling mechanisms, kept for A/B measurements (moved here from cmax_hip.h in ABI 6: they do not deliver).  Both calls
 * replace the context's OWN stream (they wait for its queued work first; not with cmx_set_stream's caller-owned stream):
 *   cmx_set_stream_priority  level > 0: the device's highest stream priority, 0: normal, < 0: lowest;
 *   cmx_set_cu_mask          the stream's kernels run on the compute units whose bit is set (hipExtStreamCreateWithCUMask;
 *                            n_words x 32 bits; which physical unit a bit selects is the driver's mapping); n_words == 0: all
 *                            compute units, normal priority.
 * Measured on MI355X (profiles/r04_fe_beside_be.txt): with both contexts evaluating back to back the default -- dynamic sharing --
 * costs the front end x1.9 and the back end x1.23; priorities change nothing (the back end's launches are one resident round of
 * workgroups that fill the register files; a queue's priority does not pre-empt them); disjoint masks isolate the two (beside =
 * solo x1.04-1.09) at the price of a static split -- e.g. front end 51.8 us (x1.34) / back end x1.82 with half of every XCD each.
 * What does work is cmx_set_sched_class (cmax_hip.h).
 * A masked stream is a BLOCKING stream (hipExtStreamCreateWithCUMask takes no flags; every other stream of this library is
 * hipStreamNonBlocking): the per-packet / per-window / per-evaluation paths issue nothing on the null stream, but the rare
 * synchronous calls that do (cmx_backend_get_map / _set_map, context creation, the device-driven solve's early-stop word) are then
 * ordered against a masked context's queue like any null-stream work.  Results never depend on either call. */
int cmx_set_stream_priority(cmx_ctx *ctx, int level);
int cmx_set_cu_mask(cmx_ctx *ctx, const uint32_t *mask, int n_words);

/* Process-wide test switches (read when a context / group is created).
 *   CMX_DIAG_FORCE_CROSS_DEVICE  1: a group whose members share ONE device is set up as if they sat on several -- the peer kernels
 *                                take their system-scope acquire (xdev) variants and the exchange's events release to system scope --
 *                                so that a one-GPU box executes the code paths of a multi-GPU group.  Results
 *                                are unchanged; the collectives get slower. */
enum { CMX_DIAG_FORCE_CROSS_DEVICE = 1 };
int cmx_diag_set(int key, int value);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#if defined(__cplusplus)
}
#endif

#endif /* CMAX_HIP_DIAG_H */
