"""How much of north_star's 1e-5 does the REFERENCE's own fp32 accumulation order use up?

The back-end fuzz sweep (tests/test_gpu_fuzz.py, same seeds) is replayed on the CPU oracle alone, twice per evaluation:
once as is, once with the events INSIDE every 100-event batch permuted (first and last event of a batch stay, so the
per-batch pose time -- hence every warped position, weight and vote -- is unchanged; only the order in which the fp32
images accumulate the votes differs).  The relative difference of the two gradients (max-norm, as tests/util.py measures
it) is the part of the HIP-vs-oracle difference that no implementation with a different vote order can avoid.

    python tests/oracle_order_noise.py [n_configs]       (CPU only; test infrastructure)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmax_slam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import backend_fuzz_config  # noqa: E402


def permute_within_batches(x, y, t, batch, rng):
    n = len(x)
    idx = np.arange(n)
    for beg in range(0, n, batch):
        end = min(beg + batch, n)
        if end - beg > 3:
            inner = idx[beg + 1:end - 1].copy()
            rng.shuffle(inner)
            idx[beg + 1:end - 1] = inner
    return x[idx], y[idx], t[idx]


def main(n_cfg):
    po.build()
    worst = []
    for seed in range(n_cfg):
        rng, k, w, IG = backend_fuzz_config(seed)
        W, H, Wp, Hp, order, K, nf, N = k["W"], k["H"], k["Wp"], k["Hp"], k["order"], k["K"], k["nf"], k["N"]
        batch, rate, sigma, measure = k["batch"], k["rate"], k["sigma"], k["measure"]
        if rate != 1 or batch < 4:
            continue   # the permutation must not change which events a sub-sampled batch keeps
        a = po.Backend(W, H, w.lut, Wp, Hp, order, batch, rate, sigma, measure)
        a.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
        px, py, pt = permute_within_batches(w.x, w.y, w.t_ns, batch, np.random.default_rng(seed))
        # timestamps travel with their events, so the old/new split of every event is unchanged; only first/last of a
        # batch define the pose time
        b = po.Backend(W, H, w.lut, Wp, Hp, order, batch, rate, sigma, measure)
        b.set_window(px, py, pt, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
        P = 3 * (K - nf)
        x = np.zeros(P)
        for step in range(5):
            rng.integers(0, 2)
            if rng.random() < 0.7:
                x = rng.normal(0, float(rng.choice([0.002, 0.02, 0.1])), P)
            ca, ga = a.eval(x, True)
            cb, gb = b.eval(x, True)
            rel = float(np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-30))
            worst.append((rel, abs(ca - cb) / abs(ca), seed, step, sigma, measure, N, P))
    worst.sort(reverse=True)
    rels = np.array([r[0] for r in worst])
    print("evaluations: %d   gradient rel. difference oracle vs order-permuted oracle:" % len(rels))
    print("  median %.2e   90%% %.2e   99%% %.2e   max %.2e   share above 1e-5: %.2f%%   above 5e-6: %.2f%%" %
          (np.median(rels), np.quantile(rels, 0.9), np.quantile(rels, 0.99), rels.max(), 100 * np.mean(rels > 1e-5),
           100 * np.mean(rels > 5e-6)))
    print("  contrast: max %.2e" % max(r[1] for r in worst))
    for r in worst[:8]:
        print("  grad %.2e contrast %.1e  seed %d step %d sigma %.1f measure %d N %d P %d" % r)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 250)
