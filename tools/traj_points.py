"""Per-point cost of the solve trajectory bench.py cycles through (front end, config 2): for every recorded point,
kernel-class times and the share of votes on the global-atomic path, evaluated (a) repeatedly at that point and
(b) interleaved with the other points (the bench's order).  Usage on the GPU box: python tools/traj_points.py [be]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmax_slam_amd import _lib, evaluator, solver, synth  # noqa: E402


def kernel_us(ev, fn, reps):
    ev.timing_enable(True)
    ev.timing_get()
    for _ in range(reps):
        fn()
    t = ev.timing_get()
    ev.timing_enable(False)
    return {k: 1e3 * v[0] / v[1] for k, v in t.items() if v[1]}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "be":
        w = synth.config3(5_000_000)
        ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, w.sigma, _lib.VARIANCE)
        pts = bench.record_trajectory(ev, np.zeros(w.P), "backend", solver.BACKEND)
    else:
        p = synth.config2(1_000_000)
        ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
        ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        pts = bench.record_trajectory(ev, np.zeros(3), "frontend", solver.FRONTEND)
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    print("rebins after the recorded solve:", ev.stats()["rebins"])
    for i, x in enumerate(pts):
        for _ in range(5):
            ev.eval(x, True)
        k = kernel_us(ev, lambda: ev.eval(x, True), 40)
        st = ev.stats()
        print("point %d  x=%s  |x|=%.3f  fallback=%.4f  %s" % (i, np.array2string(x[:3], precision=3), np.linalg.norm(x), st["fallback_frac"],
              " ".join("%s=%.1f" % kv for kv in k.items())), flush=True)
    state = {"i": 0}

    def cyc():
        ev.eval(pts[state["i"] % len(pts)], True)
        state["i"] += 1
    for _ in range(16):
        cyc()
    k = kernel_us(ev, cyc, 80)
    print("cycled           %s  rebins=%d" % (" ".join("%s=%.1f" % kv for kv in k.items()), ev.stats()["rebins"]))
    ev.close()


if __name__ == "__main__":
    main()
