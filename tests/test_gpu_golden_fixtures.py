"""-m gpu: the HIP path against the COMMITTED fixtures directly -- tests/golden/frontend_small.npz, backend_small.npz (inputs and
expected outputs as data; generator oracle/gen_golden.py).  Nothing under oracle/ is imported, built or executed here: a GPU box
that could not build the checker still has a parity check of both paths (production and reference-shaped) on both ends, both
contrast measures, both spline orders, with and without a global map.  (What the fixtures are: the restatement's outputs -- the
reference holds no vectors for the IWE path; parity vs the reference proper stays capped by that.)"""
import os

import numpy as np
import pytest

from cmax_slam_amd import _lib, synth
from util import RTOL, rel_img, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", ["production", "reference_shaped"])
@pytest.mark.parametrize("measure,ck,gk", [(_lib.VARIANCE, "contrast_var", "grad_var"), (_lib.MEAN_SQUARE, "contrast_ms", "grad_ms")])
def test_frontend_fixture(hip, path, measure, ck, gk):
    g = np.load(os.path.join(G, "frontend_small.npz"))
    W, H, fx, fy, cx, cy = int(g["W"]), int(g["H"]), float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"])
    lut = synth.pinhole_lut(W, H, fx, fy, cx, cy)
    fe = (hip if path == "production" else hip.reference_shaped).FrontendEvaluator(W, H, lut)
    fe.set_packet(g["x"], g["y"], g["t_ns"], int(g["t_ref_ns"]), fx, fy, cx, cy, 100, 1.0, measure)
    for i, om in enumerate(g["omegas"]):
        c, gr = fe.eval(om)
        assert rel_scalar(c, float(g[ck][i])) < RTOL, (i, c, float(g[ck][i]))
        assert rel_vec(gr, g[gk][i]) < RTOL, (i, gr, g[gk][i])
        c_only, _ = fe.eval(om, False)
        assert rel_scalar(c_only, float(g[ck][i])) < RTOL
        if measure == _lib.VARIANCE:
            assert rel_img(fe.computeImageOfWarpedEvents(om, blur=False), g["iwe_raw"][i]) < RTOL   # display overload: no blur
            b, d = fe.computeImageOfWarpedEvents(om, want_deriv=True, blur=True)
            assert rel_img(b, g["iwe_blur"][i]) < RTOL
            assert rel_img(d, g["deriv_blur"][i]) < RTOL                                            # interleaved H x W x 3
    fe.close()


@pytest.mark.parametrize("path", ["production", "reference_shaped"])
@pytest.mark.parametrize("tag", ["lin", "cub"])
def test_backend_fixture(hip, path, tag):
    g = np.load(os.path.join(G, "backend_small.npz"))
    v = lambda k: g[tag + "_" + k]
    W, H, Wp, Hp = int(v("W")), int(v("H")), int(v("Wp")), int(v("Hp"))
    lut = synth.pinhole_lut(W, H, float(v("fx")), float(v("fy")), float(v("cx")), float(v("cy")))
    be = (hip if path == "production" else hip.reference_shaped).BackendEvaluator(W, H, lut, Wp, Hp)
    IG = v("IG") if np.any(v("IG")) else None                      # "lin" carries a non-zero map: alpha != 0
    be.set_window(v("x"), v("y"), v("t_ns"), int(v("order")), v("knots"), int(v("start_ns")), int(v("dt_ns")), int(v("num_fixed")),
                  int(v("t_next")), 100, 1, 1.0, _lib.VARIANCE, IG)
    P = 3 * (int(v("K")) - int(v("num_fixed")))
    c0, g0 = be.eval(np.zeros(P))
    assert rel_scalar(c0, float(v("c0"))) < RTOL and rel_vec(g0, v("g0")) < RTOL
    if float(v("alpha")) != 0:
        assert IG is not None and rel_scalar(be.alpha, float(v("alpha"))) < RTOL
    else:
        assert be.alpha == 0.0
    c1, g1 = be.eval(v("drot"))
    assert rel_scalar(c1, float(v("c1"))) < RTOL and rel_vec(g1, v("g1")) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IL_OLD), v("IL_old")) < RTOL and rel_img(be.get_plane(_lib.PLANE_IL_NEW), v("IL_new")) < RTOL
    assert rel_img(be.get_plane(_lib.PLANE_IWE), v("iwe")) < RTOL
    if path == "reference_shaped":                                  # the derivative planes exist on this path only
        _, planes = be.computeImageOfWarpedEvents(v("drot"), want_deriv=True)
        assert rel_img(planes[0], v("plane_first")) < RTOL and rel_img(planes[-1], v("plane_last")) < RTOL
        # every plane's sum (a derivative plane sums to ~0: positive and negative votes cancel -- the scale is the sum of magnitudes)
        scale = np.abs(planes).sum(axis=(1, 2), dtype=np.float64).max()
        assert np.abs(planes.sum(axis=(1, 2), dtype=np.float64) - v("plane_sums")).max() < RTOL * scale
    be.close()
