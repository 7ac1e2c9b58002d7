"""-m gpu: a plain C++ host (examples/gsl_style_host.cpp: GSL-shaped f/df/fdf callbacks over the C ABI, no Python in
the loop) must reproduce the oracle's evaluation and the oracle-driven solve."""
import os
import struct
import subprocess

import numpy as np
import pytest

from cmax_slam_amd import solver, synth
from util import RTOL, rel_scalar, rel_vec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_matches_oracle(oracle, tmp_path):
    exe = os.path.join(ROOT, "examples", "gsl_style_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    p = synth.frontend_packet(40_000, 240, 180, 200.0, 200.0, 119.5, 89.5, seed=33)
    f = tmp_path / "events.bin"
    with open(f, "wb") as fh:
        fh.write(struct.pack("<iiqq4d", p.W, p.H, len(p.x), p.t_ref_ns, p.fx, p.fy, p.cx, p.cy))
        fh.write(p.x.astype("<u2").tobytes())
        fh.write(p.y.astype("<u2").tobytes())
        fh.write(p.t_ns.astype("<i8").tobytes())
        fh.write(np.ascontiguousarray(p.lut, "<f8").tobytes())
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vals = {ln.split()[0]: [float(v) for v in ln.split()[1:] if v.replace(".", "").replace("-", "").replace("e", "").replace("+", "").isdigit()]
            for ln in out.stdout.strip().splitlines()}
    ref = oracle.Frontend(p.W, p.H, p.lut, p.fx, p.fy, p.cx, p.cy, p.batch, p.sigma, 0)
    ref.set_packet(p.x, p.y, p.t_ns, p.t_ref_ns)
    c_ref, g_ref = ref.eval((0.3, -0.5, 0.2))
    assert rel_scalar(-vals["f0"][0], c_ref) < RTOL
    assert rel_vec(-np.array(vals["g0"]), g_ref) < RTOL

    def fdf(x, wg):
        c, g = ref.eval(x, wg)
        return -c, (-g if wg else None)
    x_ref, rep_ref = solver.frcg_minimize(fdf, np.zeros(3), **solver.FRONTEND)
    assert np.abs(np.array(vals["w"]) - x_ref).max() < 0.02
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("iterations")][0].split()
    assert abs(int(line[1]) - rep_ref["iterations"]) <= 2
    assert abs(float(line[-1]) - rep_ref["final_cost"]) < 1e-3 * abs(rep_ref["final_cost"])


@pytest.mark.parametrize("devices,store", [(None, False), ("0,0", False), ("0,0,0", False), ("0", True), ("0,0,0", True)])
def test_cpp_backend_window_host(oracle, tmp_path, devices, store):
    """examples/backend_window_host.cpp: one whole back-end window from C++ -- angular-velocity integration and
    control-pose fit (host fp64), window hand-over with the resident map, GSL-shaped callbacks + FR-CG, trajectory
    update, map upkeep.  devices = "0,0" / "0,0,0": the SAME host code on a group handle (cmx_backend_create_group; two / three
    members sharing this box's GPU) -- the one-process multi-GPU form the reference's single back-end thread can use.
    store: the events pushed in packets into the device event store (a replica per device of the group), the window cut from it
    (cmx_backend_set_window_from on the plain or the group handle), old events dropped afterwards -- same results."""
    exe = os.path.join(ROOT, "examples", "backend_window_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s", "backend_window_host"])
    w = synth.backend_window(30_000, 240, 180, 200.0, 200.0, 119.5, 89.5, 512, 256, 2, 5, 1, 0.2, seed=91)
    rng = np.random.default_rng(92)
    t_beg, t_end = int(w.start_ns), int(w.start_ns) + 200_000_000
    av_t = t_beg + 5_000_000 + 10_000_000 * np.arange(19, dtype=np.int64)
    av_w = np.cumsum(rng.normal(0, 0.05, (19, 3)), axis=0) + np.array([0.2, 0.8, -0.1])
    f = tmp_path / "window.bin"
    with open(f, "wb") as fh:
        fh.write(struct.pack("<8i", w.W, w.H, w.Wp, w.Hp, w.order, w.K, w.num_fixed, len(av_t)))
        fh.write(struct.pack("<6q", len(w.x), w.start_ns, w.dt_ns, w.t_next_win_beg_ns, t_beg, t_end))
        fh.write(struct.pack("<d", 0.05))
        fh.write(w.x.astype("<u2").tobytes())
        fh.write(w.y.astype("<u2").tobytes())
        fh.write(w.t_ns.astype("<i8").tobytes())
        fh.write(np.ascontiguousarray(w.lut, "<f8").tobytes())
        fh.write(np.ascontiguousarray(w.knots_init, "<f8").tobytes())
        fh.write(av_t.astype("<i8").tobytes())
        fh.write(np.ascontiguousarray(av_w, "<f8").tobytes())
    out = subprocess.run([exe, str(f)] + ([devices] if devices else []) + (["store"] if store else []), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    vals = {}
    for ln in out.stdout.strip().splitlines():
        k, *rest = ln.split()
        vals[k] = rest
    num = lambda key: np.array([float(v) for v in vals[key]])  # noqa: E731
    ref = oracle.Backend(w.W, w.H, w.lut, w.Wp, w.Hp, w.order)
    ref.set_window(w.x, w.y, w.t_ns, w.knots_init, w.start_ns, w.dt_ns, w.num_fixed, w.t_next_win_beg_ns)
    c_ref, g_ref = ref.eval(np.zeros(w.P))
    assert rel_scalar(-num("f0")[0], c_ref) < RTOL and rel_vec(-num("g0"), g_ref) < RTOL

    def fdf(x, wg):
        c, g = ref.eval(x, wg)
        return -c, (-g if wg else None)
    x_ref, rep_ref = solver.frcg_minimize(fdf, np.zeros(w.P), **solver.BACKEND)
    drotv = num("drotv")
    it = {vals["iterations"][i]: vals["iterations"][i + 1] for i in range(1, len(vals["iterations"]) - 1, 2)}
    assert abs(float(it["final"]) - rep_ref["final_cost"]) < 1e-3 * abs(rep_ref["final_cost"])
    assert abs(int(vals["iterations"][0]) - rep_ref["iterations"]) <= 2 and np.abs(drotv - x_ref).max() < 0.02
    # Trajectory::incrementalUpdate and evaluate on the host side of the ABI
    knots = num("knots").reshape(-1, 4)
    np.testing.assert_allclose(knots, oracle.left_update(w.knots_init, drotv, w.num_fixed), rtol=0, atol=1e-14)
    q_latest, _, _, _ = oracle.spline_eval(w.order, knots, w.start_ns, w.dt_ns, t_end - 1000, jac=False)
    np.testing.assert_allclose(num("latest"), q_latest, rtol=0, atol=1e-13)
    # integrateAngVel + fitCtrlPoses
    pt, pq, _, _ = oracle.integrate_ang_vel(av_t, av_w, t_beg, w.knots_init[0], int(av_t[0]), av_w[0], True)
    n_cp = oracle.num_ctrl_poses(w.order, t_beg, t_end, 0.05)
    fitted = oracle.fit_ctrl_poses(w.order, pt, pq, oracle.lib().orc_time_to_sec(t_beg), 0.05, n_cp)
    np.testing.assert_allclose(num("fitted").reshape(-1, 4), fitted, rtol=0, atol=1e-12)
    # the global map received IL_old of the last evaluation; two FOV marks over the 0.1 s stride
    map_sum, visited, marked = float(vals["map"][0]), int(vals["map"][1]), int(vals["map"][2])
    assert map_sum > 0 and visited > 0 and marked == 2
    if store:  # deleteOldEvents: the store now begins at the next window's first event
        first_kept = int(np.searchsorted(w.t_ns, w.t_next_win_beg_ns, side="left"))
        assert [int(v) for v in vals["store"]] == [first_kept, len(w.x)]
