"""Front-end solves, device-driven (CMX_OPT_CHAIN_SOLVE 1) vs host-driven (0): ms per solve, at 1M events (640x480) and 60k events
(240x180).  Under rocprofv3 --kernel-trace --stats the per-kernel averages show what the machine's step costs inside the finalize.
    python tools/chain_ab.py [n_solves]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def run(p, n_solves, label):
    fused_env = os.environ.get("CHAIN_AB_FUSED")  # "0,1": A/B of CMX_OPT_FUSED_IMAGE inside the device-driven solve
    variants = [(c, None) for c in (0, 4, 1, 0, 4, 1)] if not fused_env else [(1, int(f)) for f in fused_env.split(",")] * 2
    for chain, fused in variants:
        fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
        fe.set_option(_lib.OPT_CHAIN_SOLVE, chain)
        if fused is not None:
            fe.set_option(_lib.OPT_FUSED_IMAGE, fused)
            label_ = "%s fused=%d" % (label, fused)
        else:
            label_ = label
        fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        for _ in range(5):
            fe.setupProblemAndOptimize(np.zeros(3))
        t0 = time.perf_counter()
        for _ in range(n_solves):
            x, rep = fe.setupProblemAndOptimize(np.zeros(3))
        el = (time.perf_counter() - t0) / n_solves
        st = fe.stats()
        print("%s chain=%d: %.4f ms per solve, %d iterations, %d f + %d df, %.0f iters/s, final %.6f, slots/solve %.1f takeovers %d rebins %d fused %d redos %d"
              % (label_, chain, el * 1e3, rep["iterations"], rep["n_f"], rep["n_df"], rep["iterations"] / el, rep["final_cost"],
                 st["chain_slots"] / max(st["chain_solves"], 1), st["chain_takeovers"], st["rebins"], st["fused_evals"], st["fused_redos"]), flush=True)
        fe.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    if which in ("both", "1m"):
        run(synth.config2(), n, "1M/640x480")
    if which in ("both", "60k"):
        run(synth.frontend_packet(60_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=5), n, "60k/240x180")
