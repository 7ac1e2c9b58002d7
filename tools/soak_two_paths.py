"""Soak of the two paths on one GPU with the round-5 waiting policy: a front-end context (CMX_SCHED_URGENT, device-driven solves at a fixed
rate) beside a back-end GROUP of two members (CMX_SCHED_BACKGROUND: held -- asleep after 50 us -- while a front-end solve is on the device),
each on its own host thread, for `seconds`; every result is compared with the same call made alone beforehand.  A lost wake-up would show
as a stall (watchdog: no progress for 5 s), a race as a different number.
    python tools/soak_two_paths.py [seconds] [spin option value]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    spin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    p = synth.config2(300_000)
    w = synth.config4_slab(2, 8, 400_000)
    fe = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
    fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    be = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0, 0])
    be.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate,
                  w.sigma, _lib.VARIANCE)
    for ev in (fe, be):
        ev.set_option(_lib.OPT_SPIN_WAIT, spin)
    x_fe, r_fe = fe.setupProblemAndOptimize(np.zeros(3))
    d = np.random.default_rng(3).normal(0, 0.003, w.P)
    c_be, g_be = be.eval(d, True)
    fe.set_sched_class(_lib.SCHED_URGENT)
    be.set_sched_class(_lib.SCHED_BACKGROUND)
    stop = time.perf_counter() + seconds
    count = [0, 0]
    bad = []
    beat = [time.perf_counter(), time.perf_counter()]

    def fe_loop():
        while time.perf_counter() < stop and not bad:
            t0 = time.perf_counter()
            x, r = fe.setupProblemAndOptimize(np.zeros(3))
            if np.abs(x - x_fe).max() > 1e-4:  # (solves are not bitwise reproducible: the votes' fp32 atomics arrive in any order)
                bad.append(("front end", x, x_fe, r))
            count[0] += 1
            beat[0] = time.perf_counter()
            dt = 0.002 - (time.perf_counter() - t0)      # a packet every 2 ms (5 x the reference's rate)
            if dt > 0:
                time.sleep(dt)

    def be_loop():
        k = 0
        while time.perf_counter() < stop and not bad:
            want = k % 3 != 1
            c, g = be.eval(d, want)
            if abs(c - c_be) > 1e-7 * abs(c_be) or (want and np.abs(g - g_be).max() > 1e-5 * np.abs(g_be).max()):
                bad.append(("back end", c, c_be))
            k += 1
            count[1] += 1
            beat[1] = time.perf_counter()

    th = [threading.Thread(target=fe_loop), threading.Thread(target=be_loop)]
    for t in th:
        t.start()
    while any(t.is_alive() for t in th):
        time.sleep(0.25)
        now = time.perf_counter()
        if now < stop and (now - beat[0] > 5.0 or now - beat[1] > 5.0):
            print("STALL: front end last progress %.1f s ago, back end %.1f s ago" % (now - beat[0], now - beat[1]), flush=True)
            os._exit(2)
    for t in th:
        t.join()
    print("spin option %d, %.0f s: %d front-end solves, %d back-end group evaluations, mismatches %d" % (spin, seconds, count[0], count[1], len(bad)))
    if bad:
        print(bad[0])
        sys.exit(1)
    fe.close()
    be.close()
    print("soak ok")


if __name__ == "__main__":
    main()
