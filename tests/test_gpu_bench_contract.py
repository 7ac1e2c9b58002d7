"""-m gpu: bench.py end to end at reduced sizes -- the ONE stdout line (< 4 KB) with the fields the driver's contract names (metric,
value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline,
cpu_baseline), everything else in bench_detail.json; the nested back-end object; the N > 1 code path as a one-rank dry run
(--force-sharded); `--gpus 2` typed without a launcher on this one-GPU box; the group form on `--group-devices 0,0`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    detail = str(tmp_path / "bench_detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-out", detail, *args], capture_output=True, text=True,
                       timeout=1200, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry the ONE JSON line and nothing else: %r" % [ln[:80] for ln in lines]
    assert len(lines[0].encode()) < 4096
    return json.loads(lines[0]), json.load(open(detail))


def _check_line(d, n_gpus, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "summary", "detail"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu_fp64") and r["unit"] in ("GB/s", "TFLOP/s", "Tinstr/s") and r["peak"] > 0
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "leg_errors" not in d, d.get("leg_errors")


def _check_detail(d):
    for k in d["kernels"]:
        assert k["frac"] is None or 0 <= k["frac"] <= 1.0, k
    assert 0 < d["whole_evaluation"]["frac"] <= 1.0


def test_default_line_front_end_with_back_end_nested(tmp_path):
    line, d = _run(tmp_path, "--steps", "40", "--warmup", "5", "--events", "200000", "--steps-backend", "10", "--cpu-seconds", "0.6",
                   "--solves", "1")
    _check_line(line, 1, 40, 5)
    _check_detail(d)
    assert "config 2" in line["config"]["workload"] and line["unit"] == "events/s"
    assert abs(line["value"] - line["config"]["events_total"] / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    s = line["summary"]
    assert s["fdf_ms"] == line["ms_per_step"] and s["cmax_iters_per_s"] > 0 and s["backend"]["cmax_iters_per_s"] > 0
    assert s["backend"]["roofline"]["frac"] <= 1.0 and s["backend"]["fdf_ms"] > 0
    # ---- the detail file: what used to be the line
    assert d["cmax"]["iters_per_s"] > 0 and d["pipelined"]["ms_per_evaluation"] > 0 and d["cost_only"]["ms_per_step"] > 0
    b = d["backend"]
    assert "config 3" in b["config"]["workload"] and b["value"] > 0 and b["roofline"]["frac"] <= 1.0 and b["cpu_baseline"]["kind"] == "port"
    for src in ("host_arrays", "device_store"):
        pw = b["per_window"][src]
        assert pw["solve_ms"] > 0 and pw["pipelined"]["ms_per_window"] <= pw["sequential"]["ms_per_window"] * 1.05, pw
    shapes = b["launch_defaults"]["shapes"]
    assert len(shapes) == 8 and {s_["P"] for s_ in shapes} == {12, 15} and all(s_["fdf_ms"] > 0 and s_["iters_per_s"] > 0 for s_ in shapes)
    fb = d["frontend_beside_backend"]
    assert fb["back_to_back"]["frontend_fdf_ms"]["ratio"] > 0.9 and fb["at_100hz"]["backend_solve_ms"]["ratio"] > 0.9
    gr = d["group"]
    assert gr["group_of_2_on_one_device"]["grad_rel_vs_single"] < 1e-5 and gr["group_of_2_on_one_device"]["contrast_rel_vs_single"] < 1e-5


def test_single_process_group_line(tmp_path):
    """--group-devices 0,0: the one-process multi-GPU form end to end (two members sharing this box's GPU): config 4 through ONE
    handle, parity against the single context inside the line, the communicator's own rank count."""
    line, d = _run(tmp_path, "--gpus", "2", "--group-devices", "0,0", "--steps", "20", "--warmup", "3", "--events", "300000",
                   "--no-cpu-baseline", "--solves", "4")
    _check_line(line, 1, 20, 3)
    _check_detail(d)
    assert "config 4" in line["config"]["workload"] and line["config"]["events_total"] == 600000
    assert d["group"]["members"] == 2 and d["group"]["events_per_member"] == [300000, 300000]
    s = line["summary"]
    assert s["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and s["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    assert s["comm"]["nranks_seen"] == 2 and s["comm"]["collectives_per_step"] >= 1 and s["cmax_iters_per_s"] > 0
    assert s["one_gpu_same_workload_events_per_s"] > 0


def test_scaling_series_in_one_invocation_dry_run(tmp_path):
    """ONE invocation prints the whole weak-scaling series.  Dry run on this box's one GPU: four members sharing
    device 0 -> rows for groups of 1, 2 and 4 members on windows of 1x, 2x, 4x the per-GPU events, each with events/s per GPU, the
    exchange's share and parity against the same window on one context -- inside the < 4 KB line."""
    line, d = _run(tmp_path, "--gpus", "4", "--group-devices", "0,0,0,0", "--steps", "12", "--warmup", "2", "--events", "200000",
                   "--no-cpu-baseline", "--solves", "0")
    _check_line(line, 1, 12, 2)
    ser = line["summary"]["scaling_series"]
    assert [r["n"] for r in ser] == [1, 2, 4], ser
    for r in ser:
        assert "error" not in r and r["events_per_s_per_gpu"] > 0, r
        if r["n"] > 1:
            assert r["comm_ms"] is not None and r["comm_ms"] > 0 and r["parity_vs_1gpu"] < 1e-5, r
    assert ser[0]["comm_ms"] == 0.0 and ser[0]["parity_vs_1gpu"] is None
    full = d["scaling_series"]
    assert full[1]["devices"] == [0, 0] and full[2]["devices"] == [0, 0, 0, 0] and full[2]["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5


def test_gpus_2_as_typed_on_a_one_gpu_box(tmp_path):
    """`python3 bench.py --gpus 2` with no launcher and ONE visible device: config 4's slab runs on what is there, n_gpus = 1, an
    error field says why, exit status 0 (the driver's scaling leg must never be an rc-1 record with no line)."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible device")
    line, d = _run(tmp_path, "--gpus", "2", "--steps", "20", "--warmup", "3", "--events", "300000", "--no-cpu-baseline", "--solves", "0")
    _check_line(line, 1, 20, 3)
    assert "2 requested" in line["error"] and "config 4" in line["config"]["workload"]
    assert line["summary"]["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5


def test_gpus_n_as_typed_with_n_devices(tmp_path):
    """With >= 2 devices: the one-process group is the headline and the process-per-GPU form is self-spawned as a nested leg."""
    import torch
    n = min(torch.cuda.device_count(), 2)
    if n < 2:
        pytest.skip("needs two visible devices")
    line, d = _run(tmp_path, "--gpus", "2", "--steps", "20", "--warmup", "3", "--events", "300000", "--no-cpu-baseline", "--solves", "0")
    _check_line(line, 2, 20, 3)
    s = line["summary"]
    assert s["comm"]["nranks_seen"] == 2 and s["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5
    assert s["process_per_gpu"]["n_gpus"] == 2 and s["process_per_gpu"]["nranks_seen"] == 2 and s["process_per_gpu"]["events_per_s"] > 0


def test_sharded_code_path_as_a_one_rank_dry_run(tmp_path):
    line, d = _run(tmp_path, "--force-sharded", "--steps", "20", "--warmup", "3", "--events", "300000", "--no-cpu-baseline")
    _check_line(line, 1, 20, 3)
    _check_detail(d)
    assert "config 4" in line["config"]["workload"]
    assert d["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and d["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    c5 = d["config5"]
    assert "config 5" in c5["config"]["workload"] and c5["value"] > 0
    assert c5["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5 and c5["parity_vs_1gpu"]["contrast_rel"] < 1e-5
    assert "comm" in d and d["comm"]["collectives_per_step"] >= 2 and d["comm"]["nranks_seen"] == 1
    assert line["summary"]["config5"]["events_per_s"] > 0
