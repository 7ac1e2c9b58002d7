"""-m gpu: the configurations of the back-end sweep where the production path and the fp32 oracle differ by MORE than
north_star's 1e-5 in some build of round 2 (7 of the first 300 seeds of tests/test_gpu_fuzz.py's generator, all next to a
stationary point of a blurred image; which of them cross the bar varies with the vote order) -- arbitrated by the oracle's own sources compiled with every `float` turned into `double`
(oracle/exact_f64.c -> liboracle_f64.so; test infrastructure).

Claim checked here: at every evaluation of these configurations the production path is within 1e-5 of the EXACT value
of the reference's formula, and wherever it is further than 1e-5 from the fp32 oracle, the oracle is at least that far
from the exact value itself -- the difference is the reference arithmetic's own fp32 rounding, not the GPU path's.
(Over 300 configurations x 5 evaluations, profiles/r02d_exact_noise.txt: production path vs exact max 5.1e-6, 0 % above
1e-5; fp32 oracle vs exact max 3.0e-5, 0.73 % above 1e-5.)"""
import numpy as np
import pytest

from util import RTOL, backend_fuzz_config, backend_fuzz_points, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu
HARD_SEEDS = [21, 84, 117, 192, 211, 228, 282]


@pytest.mark.parametrize("seed", HARD_SEEDS)
def test_production_path_is_within_1e5_of_exact_arithmetic(hip, oracle, seed):
    rng, k, w, IG = backend_fuzz_config(seed)
    args = (k["W"], k["H"], w.lut, k["Wp"], k["Hp"], k["order"], k["batch"], k["rate"], k["sigma"], k["measure"])
    win = (w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, k["nf"], w.t_next_win_beg_ns, IG)
    ref = oracle.Backend(*args)
    ref.set_window(*win)
    ex = oracle.BackendExact(*args)
    ex.set_window(*win)
    be = hip.BackendEvaluator(k["W"], k["H"], w.lut, k["Wp"], k["Hp"])
    be.set_fast_path()
    be.set_window(w.x, w.y, w.t_ns, k["order"], w.knots_init, w.start_ns, w.dt_ns, k["nf"], w.t_next_win_beg_ns, k["batch"],
                  k["rate"], k["sigma"], k["measure"], IG)
    worst_vs_oracle = 0.0
    for step, (_, x) in enumerate(backend_fuzz_points(rng, k["P"])):
        c_or, g_or = ref.eval(x, True)
        c_ex, g_ex = ex.eval(x)
        c, g = be.eval(x, True)
        tag = (seed, step, k["sigma"], k["measure"])
        assert rel_scalar(c, c_ex) < RTOL and rel_scalar(c, c_or) < RTOL, tag
        assert rel_vec(g, g_ex) < RTOL, (tag, rel_vec(g, g_ex))                       # within 1e-5 of the exact value
        d_or = rel_vec(g, g_or)
        if d_or >= RTOL:                                                              # further than that from the oracle:
            assert rel_vec(g_or, g_ex) > d_or - RTOL, (tag, d_or, rel_vec(g_or, g_ex))  # the oracle's own rounding
        worst_vs_oracle = max(worst_vs_oracle, d_or)
    print("seed %d: worst production-vs-oracle difference %.2e" % (seed, worst_vs_oracle))
