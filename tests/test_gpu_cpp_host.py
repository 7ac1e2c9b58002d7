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
