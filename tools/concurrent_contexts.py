#!/usr/bin/env python
"""Aggregate throughput of several evaluator contexts on ONE GPU, each driven by its own host thread on its own stream --
the reference already runs its front end and back end on two threads (src/node.cpp:22, src/cmax_slam.cpp:92).  One
context is bound by the latency of its four dependent launches per evaluation, not by the GPU's throughput, so
independent packets (or a front-end packet and a back-end window) overlap.

  python tools/concurrent_contexts.py [--events 1000000] [--threads 1 2 4 8] [--seconds 1.0]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--with-backend", action="store_true", help="one of the contexts is a 5M-event back-end window")
    args = ap.parse_args()
    p = synth.config2(args.events)
    w = synth.config3() if args.with_backend else None
    x0 = np.array([0.3, -0.5, 0.2])
    for T in args.threads:
        evs = []
        for k in range(T):
            if w is not None and k == 0:
                ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
                ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns,
                              w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
                evs.append((ev, np.zeros(w.P), len(w.x)))
            else:
                ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
                ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
                evs.append((ev, x0, len(p.x)))
        for ev, x, _ in evs:
            ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
            for _ in range(10):
                ev.eval(x, True)
        counts = [0] * T
        stop = time.perf_counter() + args.seconds
        start = threading.Barrier(T)

        def run(k):
            ev, x, _ = evs[k]
            start.wait()
            n = 0
            while time.perf_counter() < stop:
                ev.eval(x, True)
                n += 1
            counts[k] = n

        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(k,)) for k in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        events = sum(c * e[2] for c, e in zip(counts, evs))
        print("%d context(s): %s evaluations in %.2f s -> %.3g events/s aggregate, %.1f us per evaluation per context"
              % (T, counts, el, events / el, el / max(min(counts), 1) * 1e6))
        for ev, _, _ in evs:
            ev.close()


if __name__ == "__main__":
    main()
