"""A/B timing of one evaluation in a tight host loop (no oracle, no torch): fdf and cost-only, front end (config 2) and
back end (config 3), for a list of option settings.  Usage on the GPU box:
    python tools/ab_eval.py [fe|be|both] [key=value ...]     keys: tail, spin, reuse (ints); events=N; reps=N
Prints one line per variant: ms per fdf, ms per cost-only evaluation, kernel-class times (HIP events, separate pass)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMAX_HIP_NO_TORCH", "1")
from cmax_slam_amd import _lib, evaluator, synth  # noqa: E402


def timed(fn, reps):
    for _ in range(20):
        fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


def run(kind, events, reps, variants):
    if kind == "fe":
        p = synth.config2(events or 1_000_000)
        ev = evaluator.FrontendEvaluator(p.W, p.H, p.lut)
        ev.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, _lib.VARIANCE)
        x0 = np.array([0.3, -0.5, 0.2]) if not os.environ.get("AB_OMEGA_TRUE") else np.array(p.omega_true, dtype=float)
    else:
        w = synth.config3(events or 5_000_000)
        ev = evaluator.BackendEvaluator(w.W, w.H, w.lut, w.Wp, w.Hp)
        ev.set_window(w.x, w.y, w.t_ns, w.order, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns, w.batch,
                      w.sample_rate, w.sigma, _lib.VARIANCE)
        x0 = np.zeros(w.P)
    ev.set_option(_lib.OPT_REUSE_IMAGE, 0)
    if os.environ.get("AB_BIN_AT_ZERO"):
        # the tile sort is taken at the first evaluation: make that AB_BIN_SCALE * x0 (default 0), then time at scale * x0
        ev.eval(x0 * float(os.environ.get("AB_BIN_SCALE", "0")), True)
        scale = float(os.environ["AB_BIN_AT_ZERO"])
        x0 = x0 * scale
    for name, opts in variants:
        for k, v in opts.items():
            ev.set_option(k, v)
        c, g = ev.eval(x0, True)
        ms_fdf = timed(lambda: ev.eval(x0, True), reps)
        ms_f = timed(lambda: ev.eval(x0, False), reps)
        xs = np.tile(np.asarray(x0, float), (50, 1))
        ms_fdf_c = timed(lambda: ev.eval_each(xs, True), max(reps // 50, 2)) / 50
        ms_f_c = timed(lambda: ev.eval_each(xs, False), max(reps // 50, 2)) / 50
        ev.timing_enable(True)
        ev.timing_get()
        for _ in range(50):
            ev.eval(x0, True)
        tim = ev.timing_get()
        for _ in range(50):
            ev.eval(x0, False)
        timf = ev.timing_get()
        ev.timing_enable(False)
        ks = " ".join("%s=%.1f" % (k, 1e3 * v[0] / v[1]) for k, v in tim.items() if v[1])
        kf = " ".join("%s=%.1f" % (k, 1e3 * v[0] / v[1]) for k, v in timf.items() if v[1])
        st = ev.stats()
        print("%s %-20s native loop: fdf %.4f ms  f %.4f ms" % (kind, name, ms_fdf_c, ms_f_c))
        print("%s %-20s fdf %.4f ms  f %.4f ms  c=%.10g g=%s rebins=%d fallback=%.4f  fdf kernels(us): %s   f kernels(us): %s"
              % (kind, name, ms_fdf, ms_f, c, np.array2string(np.asarray(g)[:3], precision=8), st["rebins"], st["fallback_frac"], ks, kf), flush=True)
    ev.close()


def main():
    kinds = ["fe", "be"]
    kv = {}
    for a in sys.argv[1:]:
        if a in ("fe", "be"):
            kinds = [a]
        elif a == "both":
            kinds = ["fe", "be"]
        elif "=" in a:
            k, v = a.split("=")
            kv[k] = int(v)
    variants = [("default", {})]
    if "variants" in kv:
        variants = [("tail=1", {_lib.OPT_TAIL_FINALIZE: 1}), ("tail=0", {_lib.OPT_TAIL_FINALIZE: 0})]
    if "composite" in kv:
        variants = [("composite=0", {_lib.OPT_COMPOSITE_IMAGE: 0}), ("composite=1", {_lib.OPT_COMPOSITE_IMAGE: 1}),
                    ("composite=0", {_lib.OPT_COMPOSITE_IMAGE: 0}), ("composite=1", {_lib.OPT_COMPOSITE_IMAGE: 1})]
    if "tailpoll" in kv:
        variants = [("tail tickets", {_lib.OPT_TAIL_FINALIZE: 1}), ("tail polling", {_lib.OPT_TAIL_FINALIZE: 3})] * 3
    if "fused" in kv:
        variants = [("fused=0", {_lib.OPT_FUSED_IMAGE: 0}), ("fused=1", {_lib.OPT_FUSED_IMAGE: 1}), ("fused=3", {_lib.OPT_FUSED_IMAGE: 3})] * 2
    if "fold" in kv:
        variants = [("fold=0", {_lib.OPT_FOLD_BATCH: 0}), ("fold=1", {_lib.OPT_FOLD_BATCH: 1})] * 2
    if "det" in kv:
        variants = [("deterministic=0", {_lib.OPT_DETERMINISTIC: 0}), ("deterministic=1", {_lib.OPT_DETERMINISTIC: 1})]
    for kind in kinds:
        run(kind, kv.get("events"), kv.get("reps", 300 if kind == "fe" else 100), variants)


if __name__ == "__main__":
    main()
