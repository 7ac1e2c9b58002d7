"""CPU: the oracle against the committed golden vectors.

spline_vectors.npz was produced by the REFERENCE's own vendored Basalt / Sophus (compiled from /root/reference by
oracle/Makefile `ref`; generator oracle/gen_golden.py) -> pins oracle/so3_spline.c, i.e. rows B2/B5 of SURVEY.md
section 8a.  frontend_small / backend_small are regression vectors of the oracle itself (the reference holds no
fixtures for the IWE path; parity there is unpinned at the OpenCV/ROS boundary)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def spl():
    return np.load(os.path.join(G, "spline_vectors.npz"))


@pytest.mark.parametrize("order", [2, 4])
def test_spline_value_and_jacobian_match_basalt(oracle, spl, order):
    p = "o%d_" % order
    n = len(spl[p + "K"])
    assert n >= 50
    for i in range(n):
        K = int(spl[p + "K"][i])
        q, R, J, idx = oracle.spline_eval(order, spl[p + "knots"][i][:K], spl[p + "start_ns"][i], spl[p + "dt_ns"][i],
                                          spl[p + "t_ns"][i])
        assert idx == spl[p + "idx"][i]
        np.testing.assert_allclose(q, spl[p + "quat"][i], rtol=0, atol=5e-15)
        np.testing.assert_allclose(R, spl[p + "R"][i], rtol=0, atol=5e-15)
        np.testing.assert_allclose(J, spl[p + "J"][i], rtol=0, atol=2e-14)


def test_exp_log_update_match_sophus(oracle, spl):
    for w, q, lw in zip(spl["exp_w"], spl["exp_q"], spl["log_w"]):
        np.testing.assert_allclose(oracle.so3_exp(w), q, rtol=0, atol=1e-15)
        np.testing.assert_allclose(oracle.so3_log(q), lw, rtol=0, atol=1e-15)
    out = oracle.left_update(spl["upd_knots"], spl["upd_drot"], 3)
    np.testing.assert_allclose(out, spl["upd_out"], rtol=0, atol=1e-15)


def test_live_basalt_when_built(oracle):
    """In the build container the compiled reference is present: compare on fresh random inputs too."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref/libbasalt_ref.so not built (no /root/reference on this box)")
    rng = np.random.default_rng(99)
    for order in (2, 4):
        for _ in range(100):
            K = order + int(rng.integers(0, 5))
            knots = np.array([oracle.so3_exp(rng.normal(0, 0.8, 3)) for _ in range(K)])
            dt = 50_000_000
            t = 10**9 + int(rng.integers(0, (K - order + 1) * dt))
            a = oracle.spline_eval(order, knots, 10**9, dt, t)
            b = oracle.spline_eval(order, knots, 10**9, dt, t, use_ref=True)
            assert a[3] == b[3]
            for u, v in zip(a[:3], b[:3]):
                np.testing.assert_allclose(u, v, rtol=0, atol=5e-14)


def test_frontend_regression_vectors(oracle):
    g = np.load(os.path.join(G, "frontend_small.npz"))
    from cmax_slam_amd import synth
    lut = synth.pinhole_lut(int(g["W"]), int(g["H"]), float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]))
    for measure, ck, gk in ((0, "contrast_var", "grad_var"), (1, "contrast_ms", "grad_ms")):
        fe = oracle.Frontend(int(g["W"]), int(g["H"]), lut, float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]),
                             100, 1.0, measure)
        fe.set_packet(g["x"], g["y"], g["t_ns"], int(g["t_ref_ns"]))
        for i, om in enumerate(g["omegas"]):
            c, gr = fe.eval(om)
            assert c == pytest.approx(float(g[ck][i]), rel=1e-12)
            np.testing.assert_allclose(gr, g[gk][i], rtol=1e-10, atol=1e-12)
            if measure == 0:
                np.testing.assert_array_equal(fe.iwe(om, blur=False), g["iwe_raw"][i])
                b, d = fe.iwe(om, deriv=True, blur=True)
                np.testing.assert_array_equal(b, g["iwe_blur"][i])
                np.testing.assert_array_equal(d, g["deriv_blur"][i])


@pytest.mark.parametrize("tag", ["lin", "cub"])
def test_backend_regression_vectors(oracle, tag):
    g = np.load(os.path.join(G, "backend_small.npz"))
    v = lambda k: g[tag + "_" + k]
    from cmax_slam_amd import synth
    lut = synth.pinhole_lut(int(v("W")), int(v("H")), float(v("fx")), float(v("fy")), float(v("cx")), float(v("cy")))
    be = oracle.Backend(int(v("W")), int(v("H")), lut, int(v("Wp")), int(v("Hp")), int(v("order")), 100, 1, 1.0, 0)
    IG = v("IG") if np.any(v("IG")) else None
    be.set_window(v("x"), v("y"), v("t_ns"), v("knots"), int(v("start_ns")), int(v("dt_ns")), int(v("num_fixed")),
                  int(v("t_next")), IG)
    P = 3 * (int(v("K")) - int(v("num_fixed")))
    c0, g0 = be.eval(np.zeros(P))
    assert c0 == pytest.approx(float(v("c0")), rel=1e-12)
    np.testing.assert_allclose(g0, v("g0"), rtol=1e-10, atol=1e-12)
    assert be.alpha == pytest.approx(float(v("alpha")), rel=1e-12, abs=0)
    c1, g1 = be.eval(v("drot"))
    assert c1 == pytest.approx(float(v("c1")), rel=1e-12)
    np.testing.assert_allclose(g1, v("g1"), rtol=1e-10, atol=1e-12)
    iwe, planes = be.iwe(v("drot"), planes=True)
    np.testing.assert_array_equal(iwe, v("iwe"))
    np.testing.assert_array_equal(be.IL_old, v("IL_old"))
    np.testing.assert_array_equal(planes[0], v("plane_first"))
    np.testing.assert_array_equal(planes[-1], v("plane_last"))
