"""-m gpu: bench.py end to end at reduced sizes -- the one JSON line and the fields the driver's contract names (metric, value,
unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline,
cpu_baseline), the nested back-end object, and the N > 1 code path as a one-rank dry run (--force-sharded)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip()]
    return json.loads(lines[-1])  # the JSON line is the LAST line of stdout


def _check_line(d, n_gpus, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for k in d["kernels"]:
        assert k["frac"] is None or 0 <= k["frac"] <= 1.0, k
    assert 0 < d["whole_evaluation"]["frac"] <= 1.0


def test_default_line_front_end_with_back_end_nested():
    d = _run("--steps", "40", "--warmup", "5", "--events", "200000", "--steps-backend", "10", "--cpu-seconds", "0.6", "--solves", "1")
    _check_line(d, 1, 40, 5)
    assert "config 2" in d["config"]["workload"] and d["unit"] == "events/s"
    assert abs(d["value"] - d["config"]["events_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert d["cmax"]["iters_per_s"] > 0 and d["pipelined"]["ms_per_evaluation"] > 0 and d["cost_only"]["ms_per_step"] > 0
    b = d["backend"]
    assert "config 3" in b["config"]["workload"] and b["value"] > 0 and b["roofline"]["frac"] <= 1.0 and b["cpu_baseline"]["kind"] == "port"
    # round 4: the back end's per-window pipeline, the reference's launch-default shapes, the two paths on one GPU, the group
    for src in ("host_arrays", "device_store"):
        pw = b["per_window"][src]
        assert pw["solve_ms"] > 0 and pw["pipelined"]["ms_per_window"] <= pw["sequential"]["ms_per_window"] * 1.05, pw
    shapes = b["launch_defaults"]["shapes"]
    assert len(shapes) == 8 and {s["P"] for s in shapes} == {12, 15} and all(s["fdf_ms"] > 0 and s["iters_per_s"] > 0 for s in shapes)
    fb = d["frontend_beside_backend"]
    assert fb["back_to_back"]["frontend_fdf_ms"]["ratio"] > 0.9 and fb["at_100hz"]["backend_solve_ms"]["ratio"] > 0.9
    gr = d["group"]
    assert gr["group_of_2_on_one_device"]["grad_rel_vs_single"] < 1e-5 and gr["group_of_2_on_one_device"]["contrast_rel_vs_single"] < 1e-5
    assert list(d)[-1] == "summary" and d["summary"]["fdf_ms"] == d["ms_per_step"] and d["summary"]["backend"]["cmax_iters_per_s"] > 0


def test_single_process_group_line():
    """--group-devices 0,0: the one-process multi-GPU form end to end (two members sharing this box's GPU): config 4 through ONE
    handle, parity against the single context inside the line."""
    d = _run("--group-devices", "0,0", "--steps", "20", "--warmup", "3", "--events", "300000", "--no-cpu-baseline", "--solves", "4")
    _check_line(d, 1, 20, 3)
    assert "config 4" in d["config"]["workload"] and d["config"]["events_total"] == 600000
    assert d["group"]["members"] == 2 and d["group"]["events_per_member"] == [300000, 300000]
    assert d["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and d["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    assert "comm" in d and d["comm"]["collectives_per_step"] >= 1 and d["cmax"]["iters_per_s"] > 0
    assert list(d)[-1] == "summary"


def test_sharded_code_path_as_a_one_rank_dry_run():
    d = _run("--force-sharded", "--steps", "20", "--warmup", "3", "--events", "300000", "--no-cpu-baseline")
    _check_line(d, 1, 20, 3)
    assert "config 4" in d["config"]["workload"]
    assert d["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and d["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    c5 = d["config5"]
    assert "config 5" in c5["config"]["workload"] and c5["value"] > 0
    assert c5["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and c5["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    assert "comm" in d and d["comm"]["collectives_per_step"] >= 2
