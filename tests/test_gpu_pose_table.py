"""-m gpu: the device pose-table kernels against vectors the REFERENCE's own compiled Basalt / Sophus produced.

tests/golden/spline_vectors.npz (oracle/gen_golden.py, built from /root/reference/thirdparty/basalt-headers'
So3Spline<N>::evaluate, so3_spline.h:218-274) holds 120 cases: knots, spline start / knot spacing, a query time, and the
value R, the N Jacobian blocks d_val_d_knot and the first control pose index -- what Trajectory::evaluate hands the warper
per event batch (src/backend/trajectory.cpp:86-110 / :329-355).  Until round 4 the device kernel was compared with these
only THROUGH the oracle (oracle pinned by the vectors, device compared with the oracle); this test feeds every case to the
kernel itself: a window of two events at the query time is one batch whose pose time is that time
(event_pano_warper.cpp:239-242), cmx_backend_get_pose_table reads back what be_splat / be_gather read.

Both forms of the kernel: the production form with host-precomputed knot-pair constants (K <= 16) and the generic form
(K > 16, reached by appending knots the query's segment never touches).  And the left-multiplied incremental update
(CopyAndIncrementalUpdate, trajectory.cpp:240-263) in front of it: knots exp(-d) * golden with increments d reproduce the
golden knots, so the same vectors pin the update + table pipeline."""
import os

import numpy as np
import pytest

from cmax_slam_amd import _lib

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def spl():
    return np.load(os.path.join(G, "spline_vectors.npz"))


@pytest.fixture(scope="module")
def be(hip):
    W = H = 8
    lut = np.zeros((H, W, 3))
    lut[..., 2] = 1.0
    b = hip.BackendEvaluator(W, H, lut, 64, 32)
    b.set_fast_path()
    yield b
    b.close()


def _qmul(a, b):  # (x, y, z, w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qexp(v):
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.array([0.5 * v[0], 0.5 * v[1], 0.5 * v[2], 1.0])
    s = np.sin(0.5 * th) / th
    return np.array([s * v[0], s * v[1], s * v[2], np.cos(0.5 * th)])


def _table(be, order, knots, start_ns, dt_ns, t_ns, drotv=None):
    x = np.zeros(2, np.uint16)
    t = np.array([t_ns, t_ns], np.int64)
    be.set_window(x, x, t, order, knots, int(start_ns), int(dt_ns), 0, int(t_ns) + 1, 100, 1, 1.0, _lib.VARIANCE, None)
    if drotv is not None:
        be.eval(drotv, True)
    R, J, idx, tb = be.get_pose_table()
    assert R.shape[0] == 1 and tb[0] == t_ns
    return R[0], J[0], int(idx[0])


def _check(order, spl, i, R, J, idx, atol_R):
    p = "o%d_" % order
    assert idx == int(spl[p + "idx"][i]), (i, idx)
    np.testing.assert_allclose(R, spl[p + "R"][i], rtol=0, atol=atol_R, err_msg="case %d" % i)
    # ddrot_ddrot_cp: 3 x 3N fp32, block k at columns 3k..3k+2 = (float) d_val_d_knot[k]  (trajectory.cpp:104-108 / :349-353)
    Jg = np.concatenate([spl[p + "J"][i][k] for k in range(order)], axis=1)
    np.testing.assert_allclose(J, Jg.astype(np.float32), rtol=0, atol=1.5e-7 * max(1.0, float(np.abs(Jg).max())),
                               err_msg="case %d" % i)


@pytest.mark.parametrize("order", [2, 4])
@pytest.mark.parametrize("form", ["precomputed-pairs", "generic"])
def test_pose_table_kernel_matches_basalt_vectors(be, spl, order, form):
    p = "o%d_" % order
    n = len(spl[p + "K"])
    assert n >= 50
    on_knot = 0
    for i in range(n):
        K = int(spl[p + "K"][i])
        knots = spl[p + "knots"][i][:K]
        if form == "generic":   # > 16 knots: the kernel without host-side pair constants; the extra knots are never read
            knots = np.concatenate([knots, np.tile([0.0, 0.0, 0.0, 1.0], (20 - K, 1))])
        start, dt, t = int(spl[p + "start_ns"][i]), int(spl[p + "dt_ns"][i]), int(spl[p + "t_ns"][i])
        on_knot += (t - start) % dt == 0
        R, J, idx = _table(be, order, knots, start, dt, t)
        _check(order, spl, i, R, J, idx, 2e-14)
    assert on_knot >= 1   # u = 0 cases are part of the vectors


@pytest.mark.parametrize("order", [2, 4])
def test_incremental_update_then_pose_table(be, spl, order):
    p = "o%d_" % order
    rng = np.random.default_rng(12)
    for i in range(0, len(spl[p + "K"]), 3):
        K = int(spl[p + "K"][i])
        g = spl[p + "knots"][i][:K]
        d = rng.normal(0, 0.05, (K, 3))
        knots0 = np.array([_qmul(_qexp(-d[k]), g[k]) for k in range(K)])   # exp(d) * knots0 = golden knots
        R, J, idx = _table(be, order, knots0, int(spl[p + "start_ns"][i]), int(spl[p + "dt_ns"][i]), int(spl[p + "t_ns"][i]),
                           drotv=d.reshape(-1))
        _check(order, spl, i, R, J, idx, 5e-13)
