"""Timeline of ONE fused splat + image launch (CMX_OPT_FUSED_IMAGE): per-workgroup wall-clock stamps written by the kernel when the
environment variable CMX_FUSE_TRACE names an output file.  Usage on the GPU box:  python tools/fuse_trace.py [events]
Prints, in microseconds from the first workgroup's start: when chunk workgroups start / finish, when tile workgroups start, see
their inputs complete and finish."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
path = "/tmp/fuse_trace.bin"
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def pct(v, name):
    v = np.sort(v)
    print("  %-34s n=%4d  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (name, len(v), v[0], v[len(v) // 10], v[len(v) // 2], v[(9 * len(v)) // 10], v[-1]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    p = synth.config2(n)
    ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
    if os.environ.get("TRACE_FUSED"):
        ev.set_option(_lib.OPT_FUSED_IMAGE, int(os.environ["TRACE_FUSED"]))
    ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
    x0 = np.array([0.3, -0.5, 0.2])
    if len(sys.argv) > 2:  # sort at omega = 0, evaluate at argv[2] x the packet's true rate (a cold-start solve's range)
        ev.eval(np.zeros(3), True)
        x0 = float(sys.argv[2]) * np.array(p.omega_true, dtype=float)
    for _ in range(30):
        ev.eval(x0 + 1e-4 * np.random.randn(3), True)
    os.environ["CMX_FUSE_TRACE"] = path  # (the library looks the variable up during its first few hundred evaluations only)
    for rep in range(2):
        ev.eval(x0 + 1e-4 * np.random.randn(3), True)
        t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
        role = t[:, 3]
        live = t[:, 0] > 0
        t0 = t[live, 0].min()
        us = lambda col, m: (t[m, col] - t0) / 100.0  # noqa: E731  (100 MHz wall clock)
        ch, tl, idle = live & (role == 1), live & (role == 2), live & (role == 3)
        ga = live & (role == 4)
        print("rebins %d fallback_frac %.5f" % (ev.stats()["rebins"], ev.stats()["fallback_frac"]))
        print("workgroups %d, by role:" % len(t), {int(r): int((role == r).sum()) for r in np.unique(role)}, "one-launch evaluations:", ev.stats()["one_launch_evals"])
        print("launch %d: %d chunk workgroups, %d strip workgroups (+ %d of empty tiles), %d workgroups beyond the table" % (rep, ch.sum(), tl.sum(), idle.sum(), (live & (role == 0)).sum()))
        pct(us(0, ch), "chunk start")
        pct(us(1, ch), "chunk flushed + arrived")
        if tl.sum() and (t[tl, 2] > 0).all():
            pct(us(0, tl), "tile start")
            pct(us(1, tl), "tile inputs complete (poll ok)")
            pct(us(2, tl), "tile end")
            pct(us(2, tl) - us(1, tl), "tile pass duration")
            pct(us(4, tl) - us(1, tl), "  raw pixels loaded (sc1)")
            pct(us(5, tl) - us(4, tl), "  LDS write + row pass")
            pct(us(6, tl) - us(5, tl), "  column pass (Jt stores issued)")
            pct(us(2, tl) - us(6, tl), "  moments + stores drained")
            if (t[tl, 7] > 0).all():
                pct(us(7, tl) - us(2, tl), "  the whole pass AGAIN (warm code)")
            pct(us(1, tl) - us(1, ch).max(), "poll ok - last chunk arrival")
        if ev.stats()["self_serve_evals"]:  # self-service form: every workgroup is a chunk workgroup (cmx_selfserve.hpp's stamps)
            dbg64 = bool(int(os.environ.get("CMX_FUSE_DEBUG", "0")) & 64)
            own = ch & (t[:, 2] > 0) & (not dbg64)
            if dbg64:
                own = ch & False
                pct(us(5, ch), "events re-read + warped")
                pct(us(6, ch), "its tiles' passes are done")
                pct(us(7, ch), "Jt cells read, sums added")
                pct(us(2, ch) - us(6, ch), "  wait ended -> first cells loaded")
                pct(us(4, ch) - us(2, ch), "  -> per-thread sums formed")
                pct(us(7, ch) - us(4, ch), "  -> wave sums, LDS, atomics issued")
                continue
            pct(us(2, own), "owner: tile inputs complete")
            pct(us(2, own) - us(1, own), "  (since the owner's own arrival)")
            pct(us(2, own) - us(1, ch).max(), "  (since the LAST chunk arrival)")
            pct(us(4, own), "owner: pass stored + published")
            pct(us(4, own) - us(2, own), "owner: pass duration")
            pct(us(5, ch), "events re-read + warped")
            pct(us(5, ch & ~own) - us(1, ch & ~own), "  (non-owners: since arrival)")
            pct(us(5, own) - us(4, own), "  (owners: since their pass)")
            pct(us(6, ch), "its tiles' passes are done")
            pct(us(7, ch), "Jt cells read, sums added")
            pct(us(7, ch) - us(6, ch), "  (since the wait ended)")
            if int(os.environ.get("CMX_FUSE_DEBUG", "0")) & 64:  # (stamps 2 and 4 re-used: cells loaded, sums formed)
                pct(us(2, ch) - us(6, ch), "  wait ended -> first cells loaded")
                pct(us(4, ch) - us(2, ch), "  -> per-thread sums formed")
                pct(us(7, ch) - us(4, ch), "  -> wave sums, LDS, atomics issued")
        if ga.sum():
            pct(us(0, ga), "gather start")
            pct(us(1, ga), "gather: events warped")
            pct(us(2, ga), "gather: its tiles are done")
            pct(us(4, ga), "gather: Jt cells read, summed")
            pct(us(5, ga), "gather: arrived")
            fin = ga & (t[:, 6] > 0)
            if fin.sum():
                pct(us(6, fin), "finalize done (last arriver)")
    ev.close()


if __name__ == "__main__":
    main()
