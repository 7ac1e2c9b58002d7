"""Soak of the fused front-end evaluation (image pass inside the splat launch, CMX_OPT_FUSED_IMAGE 1) under what could break its
waits: N host threads, each with its own front-end context (their launches share the GPU's workgroup slots, so tile workgroups of one
context wait while chunk workgroups of others are being dispatched), mixing plain fused evaluations, device-driven solves with fused
slots, cost-only evaluations, occasional jumps beyond the tiles' reach and re-sorts; a back-end context hammers large launches beside
them.  Every fused evaluation is compared with the same evaluation through the three separate launches on a reference context.
A protocol fault would show as a wrong number, a tile that gave up waiting (kFuseIncomplete -> counted in fused_redos without a
jump), or a hang.
    python tools/soak_fused.py [seconds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(nthreads))   # (bisecting: run these workers only)
    packets = [synth.frontend_packet(n, W, H, f, f, (W - 1) / 2, (H - 1) / 2, seed=300 + i)
               for i, (n, W, H, f) in enumerate([(200_000, 640, 480, 590.0), (60_000, 240, 180, 200.0), (1_000_000, 640, 480, 590.0),
                                                 (30_000, 346, 260, 300.0), (120_000, 320, 240, 250.0), (500_000, 640, 480, 590.0)])]
    w = synth.config4_slab(0, 8, 2_000_000)
    stop = threading.Event()
    no = set(os.environ.get("SOAK_NO", "").split(","))   # bisecting: any of fused, solve, jump, cost, be
    errors, counts = [], [None] * nthreads

    def backend_noise():
        be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate,
                      w.sigma, _lib.VARIANCE)
        rng = np.random.default_rng(1)
        while not stop.is_set():
            if "be" in no:
                time.sleep(0.1)
                continue
            be.eval(rng.normal(0, 0.003, w.P), True)
        be.close()

    def worker(k):
        try:
            p = packets[k % len(packets)]
            rng = np.random.default_rng(100 + k)
            fused = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
            plain = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
            plain.set_option(_lib.OPT_FUSED_IMAGE, 0)
            if "fused" in no:
                fused.set_option(_lib.OPT_FUSED_IMAGE, 0)
            for fe in (fused, plain):
                fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
            om = np.array(p.omega_true, float) * 0.8
            n_eval = n_solve = n_jump = 0
            worst = 0.0
            hist = []
            while not stop.is_set():
                r = rng.random()
                if r < 0.02 and "jump" in no or 0.02 <= r < 0.05 and "solve" in no or 0.05 <= r < 0.15 and "cost" in no:
                    r = 1.0
                if r < 0.02:
                    om = rng.normal(0, 4.0, 3)          # beyond the reach: repeated after a fresh sort
                    n_jump += 1
                    hist.append("jump")
                elif r < 0.05:
                    x, rep = fused.setupProblemAndOptimize(om + rng.normal(0, 0.05, 3))   # device-driven solve, fused slots
                    x2, rep2 = plain.setupProblemAndOptimize(x)
                    n_solve += 1
                    hist.append("solve it=%d/%d" % (rep["iterations"], rep2["iterations"]))
                    continue
                elif r < 0.15:
                    c = fused.eval(om, False)[0]
                    c2 = plain.eval(om, False)[0]
                    worst = max(worst, abs(c - c2) / max(abs(c2), 1e-30))
                    hist.append("cost")
                    continue
                else:
                    om = om + rng.normal(0, 0.02, 3)
                    hist.append("step")
                c, g = fused.eval(om, True)
                c2, g2 = plain.eval(om, True)
                n_eval += 1
                ec = abs(c - c2) / max(abs(c2), 1e-30)
                eg = float(np.abs(g - g2).max() / max(np.abs(g2).max(), 1e-30))
                worst = max(worst, ec, eg)
                if ec > 1e-6 or eg > 1e-5:
                    fresh = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
                    fresh.set_option(_lib.OPT_FUSED_IMAGE, 0)
                    fresh.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
                    c3, g3 = fresh.eval(om, True)
                    c4, g4 = fused.eval(om + 1e-9, True)
                    c5, g5 = plain.eval(om + 1e-9, True)
                    i0 = fresh.computeImageOfWarpedEvents(om, blur=False)
                    b0 = fresh.computeImageOfWarpedEvents(om, blur=True)
                    for name, ctx in (("plain", plain), ("fused", fused)):
                        i1 = ctx.computeImageOfWarpedEvents(om, blur=False)
                        b1 = ctx.computeImageOfWarpedEvents(om, blur=True)
                        d = np.abs(i1 - i0) > 1e-3
                        ys, xs = np.nonzero(d)
                        print("IMAGES %s sum %.3f fresh sum %.3f; %d pixels differ, bbox x %s y %s, sum of differences %.3f; blurred: max diff %.3g" %
                              (name, i1.sum(), i0.sum(), d.sum(), (xs.min(), xs.max()) if d.any() else None, (ys.min(), ys.max()) if d.any() else None,
                               float((i1 - i0)[d].sum()), float(np.abs(b1 - b0).max())), flush=True)
                    c8, g8 = fused.eval(om + 2e-9, True)
                    fused.prepare(om)
                    c9, g9 = fused.eval(om + 3e-9, True)
                    print("fused after the images %r, after a fresh sort %r" % (c8, c9), flush=True)
                    c6, g6 = plain.eval(om + 2e-9, True)
                    plain.prepare(om)
                    c7, g7 = plain.eval(om + 3e-9, True)
                    print("plain after the images %r, after a fresh sort %r; plain stats %r" % (c6, c7, plain.stats()), flush=True)
                    print("MISMATCH thread %d after %r\n fresh %r %r\n fused again %r plain again %r\n stats %r" %
                          (k, hist[-12:], c3, g3, c4, c5, fused.stats()), flush=True)
                    raise AssertionError("thread %d: fused %r %r vs separate %r %r at %r" % (k, c, g, c2, g2, om))
            st = fused.stats()
            counts[k] = (n_eval, n_solve, n_jump, worst, st["fused_evals"], st["fused_redos"], st["rebins"], st["chain_takeovers"], st["fused_timeouts"])
            fused.close()
            plain.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            stop.set()

    threads = [threading.Thread(target=worker, args=(k,)) for k in only] + [threading.Thread(target=backend_noise)]
    t0 = time.time()
    for t in threads:
        t.start()
    while time.time() - t0 < seconds and not stop.is_set():
        time.sleep(0.2)
    stop.set()
    for t in threads:
        t.join()
    if errors:
        print("SOAK FAILED:", errors)
        sys.exit(1)
    counts = [c for c in counts if c is not None]
    tot_eval = sum(c[0] for c in counts)
    tot_fused = sum(c[4] for c in counts)
    print("soak ok: %d threads + a back-end context, %.0f s: %d checked fused evaluations, %d solves, %d jumps; fused launches %d, repeated %d "
          "(jumps + solves crossing the reach), re-sorts %d, chain take-overs %d, TIMEOUTS %d; worst difference to the separate launches %.2e"
          % (nthreads, time.time() - t0, tot_eval, sum(c[1] for c in counts), sum(c[2] for c in counts), tot_fused, sum(c[5] for c in counts),
             sum(c[6] for c in counts), sum(c[7] for c in counts), sum(c[8] for c in counts), max(c[3] for c in counts)))
    for k, c in enumerate(counts):
        print("  thread %d: evals %d solves %d jumps %d worst %.2e fused %d redos %d rebins %d takeovers %d timeouts %d" % ((k,) + c))


if __name__ == "__main__":
    main()
