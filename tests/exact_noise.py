"""fp32-vs-exact error of the reference's own arithmetic, and (on a GPU box) of the HIP paths, on the back-end fuzz sweep.

`exact` = oracle/liboracle_f64.so: the CPU oracle's sources compiled with every `float` turned into `double` (images,
weights, blur taps, Jacobian chain, accumulators; oracle/exact_f64.c): the same algorithm without fp32 rounding.  Reported, as tests/util.py measures it
(max-norm error relative to the gradient's max-norm):
    oracle (fp32, = the reference's arithmetic)  vs exact
    HIP production path / reference-shaped path  vs exact  and vs the oracle          (only with --gpu)
TEST INFRASTRUCTURE (uses oracle/); results are quoted in DESIGN.md section 2.

    python tests/exact_noise.py [n_configs] [--gpu] [--frontend]      (--frontend: the same for tests/test_gpu_fuzz.py's front-end sweep)
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

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main(n_cfg, gpu):
    po.build()
    if gpu:
        from cmax_slam_amd import evaluator
    rows = []
    for seed in range(n_cfg):
        rng, k, w, IG = backend_fuzz_config(seed)
        W, H, Wp, Hp, order, K, nf, N = k["W"], k["H"], k["Wp"], k["Hp"], k["order"], k["K"], k["nf"], k["N"]
        batch, rate, sigma, measure = k["batch"], k["rate"], k["sigma"], k["measure"]
        ref = po.Backend(W, H, w.lut, Wp, Hp, order, batch, rate, sigma, measure)
        ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
        ex = po.BackendExact(W, H, w.lut, Wp, Hp, order, batch, rate, sigma, measure)
        ex.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, IG)
        hips = []
        if gpu:
            for fast in (True, False):
                be = evaluator.BackendEvaluator(W, H, w.lut, Wp, Hp)
                (be.set_fast_path if fast else be.set_reference_path)()
                be.set_window(w.x, w.y, w.t_ns, order, w.knots_init, w.start_ns, w.dt_ns, nf, w.t_next_win_beg_ns, batch, rate,
                              sigma, measure, IG)
                hips.append(be)
        P = 3 * (K - nf)
        x = np.zeros(P)
        for step in range(5):
            rng.integers(0, 2)
            if rng.random() < 0.7:
                x = rng.normal(0, float(rng.choice([0.002, 0.02, 0.1])), P)
            co, go = ref.eval(x, True)
            ce, ge = ex.eval(x)
            row = dict(seed=seed, step=step, sigma=sigma, measure=measure, N=N, P=P, oracle_vs_exact=rel(go, ge),
                       c_oracle_vs_exact=abs(co - ce) / abs(ce))
            for name, be in zip(("fast", "planes"), hips):
                c, g = be.eval(x, True)
                row[name + "_vs_exact"] = rel(g, ge)
                row[name + "_vs_oracle"] = rel(g, go)
                row["c_" + name + "_vs_exact"] = abs(c - ce) / abs(ce)
            rows.append(row)
        for be in hips:
            be.close()
    keys = [k for k in rows[0] if k.endswith("_vs_exact") or k.endswith("_vs_oracle")]
    print("evaluations: %d (%d configurations)" % (len(rows), n_cfg))
    for k in keys:
        v = np.array([r[k] for r in rows])
        print("  %-22s median %.2e  90%% %.2e  99%% %.2e  max %.2e  above 1e-5: %.2f%%" %
              (k, np.median(v), np.quantile(v, 0.9), np.quantile(v, 0.99), v.max(), 100 * np.mean(v > 1e-5)))
    for sg in sorted(set(r["sigma"] for r in rows)):
        sel = [r for r in rows if r["sigma"] == sg]
        print("  sigma %.1f (%d evaluations): max " % (sg, len(sel)) + "  ".join("%s %.2e" % (k, max(r[k] for r in sel)) for k in keys if not k.startswith("c_")))
    key = "fast_vs_oracle" if gpu else "oracle_vs_exact"
    for r in sorted(rows, key=lambda r: -r[key])[:10]:
        print("  " + "  ".join("%s=%s" % (k, ("%.2e" % v) if isinstance(v, float) and k not in ("sigma",) else v) for k, v in r.items()))


def main_frontend(n_cfg, gpu):
    """Front-end sweep: the configurations and evaluation sequences of test_frontend_random_configuration."""
    po.build()
    if gpu:
        from cmax_slam_amd import _lib, evaluator
    rows = []
    for seed in range(n_cfg):
        rng = np.random.default_rng(1000 + seed)
        W, H = int(rng.integers(48, 400)), int(rng.integers(40, 300))
        f = float(rng.uniform(0.6, 1.4) * max(W, H))
        N = int(rng.integers(1, 30_000))
        batch = int(rng.choice([1, 7, 64, 100, 257]))
        sigma = float(rng.choice([0.0, 0.5, 1.0, 1.7, 3.0]))
        measure = int(rng.choice([0, 1]))
        p = synth.frontend_packet(N, W, H, f, f, (W - 1) / 2, (H - 1) / 2, T=float(rng.uniform(0.01, 0.08)), seed=seed)
        ref = po.Frontend(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, batch, sigma, measure)
        ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        ex = po.FrontendExact(W, H, p.lut, p.fx, p.fy, p.cx, p.cy, batch, sigma, measure)
        ex.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
        hips = []
        if gpu:
            for fast in (True, False):
                fe = evaluator.FrontendEvaluator(W, H, p.lut)
                (fe.set_fast_path if fast else fe.set_reference_path)()
                fe.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, batch, sigma, measure)
                hips.append(fe)
        x = np.zeros(3)
        for step in range(5):
            rng.integers(0, 2)
            if rng.random() < 0.7:
                x = p.omega_true * rng.uniform(0, 1.5) + rng.normal(0, 0.3, 3)
            co, go = ref.eval(x, True)
            ce, ge = ex.eval(x)
            if not np.isfinite(ce) or abs(ce) < 1e-300 or np.abs(ge).max() < 1e-300:
                continue  # (a packet with no accepted vote)
            row = dict(seed=seed, step=step, sigma=sigma, measure=measure, N=N, oracle_vs_exact=rel(go, ge),
                       c_oracle_vs_exact=abs(co - ce) / abs(ce))
            for name, fe in zip(("fast", "planes"), hips):
                c, g = fe.eval(x, True)
                row[name + "_vs_exact"] = rel(g, ge)
                row[name + "_vs_oracle"] = rel(g, go)
                row["c_" + name + "_vs_exact"] = abs(c - ce) / abs(ce)
            rows.append(row)
        for fe in hips:
            fe.close()
    keys = [k for k in rows[0] if k.endswith("_vs_exact") or k.endswith("_vs_oracle")]
    print("front end -- evaluations: %d (%d configurations)" % (len(rows), n_cfg))
    for k in keys:
        v = np.array([r[k] for r in rows])
        print("  %-22s median %.2e  90%% %.2e  99%% %.2e  max %.2e  above 1e-5: %.2f%%" %
              (k, np.median(v), np.quantile(v, 0.9), np.quantile(v, 0.99), v.max(), 100 * np.mean(v > 1e-5)))
    for sg in sorted(set(r["sigma"] for r in rows)):
        sel = [r for r in rows if r["sigma"] == sg]
        print("  sigma %.1f (%d evaluations): max " % (sg, len(sel)) + "  ".join("%s %.2e" % (k, max(r[k] for r in sel)) for k in keys if not k.startswith("c_")))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    (main_frontend if "--frontend" in sys.argv else main)(int(args[0]) if args else 250, "--gpu" in sys.argv)
