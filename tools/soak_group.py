"""Soak of the one-process group's exchange (round 5: one-shot peer sum fused into the unpack, alternating send buffers, one host
barrier per collective): many evaluations with wandering and jumping parameters on groups of 2 / 3 / 4 members sharing ONE device,
every evaluation checked against a single context on the same window -- a protocol race would show as a hang (barrier timeout =
CMX_ERR_HIP after 20 s), a CMX_ERR_STATE ("members disagree") or a numerical difference.
    python tools/soak_group.py [evaluations per group size]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def main():
    n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    if len(sys.argv) > 2 and sys.argv[2] == "xdev":  # the multi-device code paths on this one device (cmax_hip_diag.h)
        assert _lib.lib().cmx_diag_set(_lib.DIAG_FORCE_CROSS_DEVICE, 1) == 0
        print("CMX_DIAG_FORCE_CROSS_DEVICE on: system-scope acquire variants of the peer kernels, release-to-system events", flush=True)
    w = synth.config4_slab(2, 8, 600_000)
    rng = np.random.default_rng(11)
    one = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
    args = (w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch, w.sample_rate, w.sigma, _lib.VARIANCE)
    one.set_window(*args)
    for members in (2, 3, 4):
        grp = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp, devices=[0] * members)
        grp.set_window(*args)
        one.set_window(*args)
        x = np.zeros(w.P)
        worst_c = worst_g = 0.0
        misses0 = grp.stats()["exchange_misses"]
        t0 = time.perf_counter()
        for k in range(n_eval):
            r = rng.random()
            if r < 0.02:
                x = np.tile(rng.normal(0, 0.3, 3), w.P // 3)       # a jump: the votes leave the exchange set (second one-shot exchange)
            elif r < 0.10:
                x = np.zeros(w.P)
            else:
                x = x + rng.normal(0, 0.002, w.P)
            want = rng.random() < 0.7
            if k % 500 == 250:                                      # a new window now and then (whole-plane exchange on its first evaluation)
                n = int(rng.integers(300_000, len(w.x)))
                a2 = (w.x[:n], w.y[:n], w.t_ns[:n]) + args[3:]
                grp.set_window(*a2)
                one.set_window(*a2)
            c, g = grp.eval(x, want)
            if k % 7 == 0:
                c1, g1 = one.eval(x, want)
                dc = abs(c - c1) / abs(c1)
                dg = float(np.abs(g - g1).max() / np.abs(g1).max()) if want else 0.0
                if dc > 1e-6 or dg > 1e-5:
                    print("  k=%d members=%d: contrast %.12g vs %.12g (rel %.2e), gradient rel %.2e, |g|max %.3e, |x|max %.3f"
                          % (k, members, c, c1, dc, dg, float(np.abs(g1).max()) if want else 0.0, float(np.abs(x).max())), flush=True)
                worst_c, worst_g = max(worst_c, dc), max(worst_g, dg)
        el = time.perf_counter() - t0
        st = grp.stats()
        print("members %d: %d evaluations in %.1f s, worst contrast rel %.2e, worst gradient rel %.2e, exchange misses %d, host syncs %d"
              % (members, n_eval, el, worst_c, worst_g, st["exchange_misses"] - misses0, st["sharded_host_syncs"]), flush=True)
        assert worst_c < 1e-6 and worst_g < 1e-5, "group and single context disagree"
        grp.close()
    one.close()
    print("soak ok")


if __name__ == "__main__":
    main()
