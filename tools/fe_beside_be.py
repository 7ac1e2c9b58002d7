#!/usr/bin/env python
"""The reference's two threads on ONE GPU: a front-end context (1M-event packets, fdf evaluations waited for one by one)
beside a back-end context (config 3 window, fdf evaluations in a loop) -- src/node.cpp:22 + src/cmax_slam.cpp:92.
Sweeps what the host can do about their interference: stream priority (cmx_set_stream_priority) and compute-unit
partitioning (cmx_set_cu_mask).

  python tools/fe_beside_be.py [--seconds 0.6] [--fe-events 1000000]
prints one line per setting: front-end us / evaluation (solo and beside), back-end us / evaluation (solo and beside).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def loop_pair(fe, be, xf, xb, seconds, fe_only=False, be_only=False):
    counts = [0, 0]
    stop = [0.0]
    active = [not be_only, not fe_only]
    go = threading.Barrier(sum(active) + 1)

    def run(k):
        ev, x = (fe, xf) if k == 0 else (be, xb)
        go.wait()
        n = 0
        while time.perf_counter() < stop[0]:
            ev.eval(x, True)
            n += 1
        counts[k] = n
    th = [threading.Thread(target=run, args=(k,)) for k in range(2) if active[k]]
    for t in th:
        t.start()
    stop[0] = time.perf_counter() + seconds + 0.02
    t0 = time.perf_counter()
    go.wait()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    return [el / c * 1e6 if c else None for c in counts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=0.6)
    ap.add_argument("--fe-events", type=int, default=1_000_000)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    p = synth.config2(args.fe_events)
    w = synth.config3()
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                  w.sample_rate, w.sigma, _lib.VARIANCE)
    xf, xb = np.array([0.3, -0.5, 0.2]), np.zeros(w.P)
    for ev, x in ((fe, xf), (be, xb)):
        ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
        for _ in range(20):
            ev.eval(x, True)

    def setting(name, fe_prio=0, be_prio=0, fe_cus=None, be_cus=None, fe_first=0, be_first=0):
        if fe_cus:
            fe.set_cu_mask(fe_cus, fe_first)
        else:
            fe.set_stream_priority(fe_prio)
        if be_cus:
            be.set_cu_mask(be_cus, be_first)
        else:
            be.set_stream_priority(be_prio)
        for _ in range(5):
            fe.eval(xf, True)
            be.eval(xb, True)
        fs = loop_pair(fe, be, xf, xb, args.seconds / 2, fe_only=True)[0]
        bs = loop_pair(fe, be, xf, xb, args.seconds / 2, be_only=True)[1]
        fb, bb = loop_pair(fe, be, xf, xb, args.seconds)
        r = {"setting": name, "fe_solo_us": fs, "be_solo_us": bs, "fe_beside_us": fb, "be_beside_us": bb}
        print("%-46s FE solo %6.1f  beside %6.1f (x%.2f)   BE solo %6.1f  beside %6.1f (x%.2f)"
              % (name, fs, fb, fb / fs, bs, bb, bb / bs), flush=True)
        return r
    def per_xcd(lo, hi):  # 32-bit words, one per XCD (measured: 32 consecutive mask bits = the 32 compute units of one XCD)
        return [(((1 << hi) - 1) ^ ((1 << lo) - 1)) & 0xffffffff] * 8

    def spread(name, k, be_all=False):  # the front end owns k compute units of EVERY XCD, the back end the other 32 - k
        fe.set_cu_mask(mask_words=per_xcd(0, k))
        if be_all:
            be.set_stream_priority(0)
        else:
            be.set_cu_mask(mask_words=per_xcd(k, 32))
        for _ in range(5):
            fe.eval(xf, True)
            be.eval(xb, True)
        fs = loop_pair(fe, be, xf, xb, args.seconds / 2, fe_only=True)[0]
        bs = loop_pair(fe, be, xf, xb, args.seconds / 2, be_only=True)[1]
        fb, bb = loop_pair(fe, be, xf, xb, args.seconds)
        print("%-46s FE solo %6.1f  beside %6.1f (x%.2f)   BE solo %6.1f  beside %6.1f (x%.2f)"
              % (name, fs, fb, fb / fs, bs, bb, bb / bs), flush=True)
        return {"setting": name, "fe_solo_us": fs, "be_solo_us": bs, "fe_beside_us": fb, "be_beside_us": bb}
    out = [setting("default streams")]
    for k in (4, 6, 8, 12, 16):
        out.append(spread("per-XCD mask: FE %d of 32 CUs | BE %d" % (k, 32 - k), k))
    for k in (8, 12):
        out.append(spread("per-XCD mask: FE %d of 32 CUs | BE all CUs" % k, k, be_all=True))
    out.append(setting("FE high priority", fe_prio=1))
    out.append(setting("FE high, BE low priority", fe_prio=1, be_prio=-1))
    out.append(setting("BE low priority", be_prio=-1))
    for k in (16, 32, 48, 64, 96):
        out.append(setting("CU mask: FE %d CUs | BE %d CUs" % (k, 256 - k), fe_cus=k, be_cus=256 - k, be_first=k))
    for k in (32, 64):
        out.append(setting("CU mask: FE %d CUs | BE all, low priority" % k, fe_cus=k, be_prio=-1))
        out.append(setting("CU mask: BE %d CUs only | FE high priority" % (256 - k), fe_prio=1, be_cus=256 - k, be_first=k))
    # reference for the ratios: the default solo times
    base_f, base_b = out[0]["fe_solo_us"], out[0]["be_solo_us"]
    for r in out:
        r["fe_vs_default_solo"] = r["fe_beside_us"] / base_f
        r["be_vs_default_solo"] = r["be_beside_us"] / base_b
    print("\nrelative to the DEFAULT solo times (FE %.1f us, BE %.1f us):" % (base_f, base_b))
    for r in out:
        print("%-46s FE x%.2f   BE x%.2f" % (r["setting"], r["fe_vs_default_solo"], r["be_vs_default_solo"]))
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)
    fe.close()
    be.close()


if __name__ == "__main__":
    main()
