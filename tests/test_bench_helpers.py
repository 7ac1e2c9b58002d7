"""CPU: bench.py's bookkeeping that does not need a GPU -- the PMC byte counts are only reported for the build they were measured
on (source hash of cmax_slam_amd/csrc stamped into profiles/pmc_traffic.json), the byte models are self-consistent."""
import json
import os
import shutil

import bench


def test_pmc_traffic_is_only_reported_for_the_build_it_was_measured_on(tmp_path, monkeypatch):
    h = bench.csrc_hash()
    assert len(h) == 64 and h == bench.csrc_hash()
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    shutil.copytree(os.path.join(bench.ROOT, "cmax_slam_amd", "csrc"), root / "cmax_slam_amd" / "csrc",
                    ignore=shutil.ignore_patterns("build"))
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.csrc_hash() == h                                   # build products are not part of the hash
    pmc, note = bench.load_pmc()
    assert pmc == {} and "missing" in note
    json.dump({"_stamp": {"src_sha256": h, "git_head": "abc"}, "_source": "x", "frontend_fast_gather": 123.0},
              open(root / "profiles" / "pmc_traffic.json", "w"))
    pmc, note = bench.load_pmc()
    assert pmc["frontend_fast_gather"] == 123.0 and "this build" in note
    with open(root / "cmax_slam_amd" / "csrc" / "cmx_warp.hpp", "a") as f:
        f.write("// touched\n")
    pmc, note = bench.load_pmc()                                     # any source change: the counts describe another build
    assert pmc == {} and "another build" in note


def test_byte_models_mandatory_never_exceeds_algorithmic_for_the_per_event_kernels():
    fe = bench.byte_models("frontend", 0, 1_000_000, 640 * 480, 10_000, 3, True, 200_000)
    be = bench.byte_models("backend", 4, 5_000_000, 1024 * 1024, 50_000, 21, True, 60_000, image_pixels=80_000)
    for m in (fe, be):
        for k in ("splat", "gather"):
            alg, mand = m[k]
            assert 0 < mand <= alg, (k, alg, mand)
    assert bench.whole_eval_bytes_8d("frontend", 0, 1_000_000, 640 * 480, 3) == 1_000_000 * 156 + 4 * 640 * 480 * 24


# ---------------------------------------------------------------- round 5: the stdout line and the launcher logic
def _recorded_detail():
    """A full round-4 result (21 KB: the line the driver could not parse) -- the input the compact line must survive."""
    return json.load(open(os.path.join(bench.ROOT, "profiles", "r04_bench.json")))


def test_stdout_line_is_small_and_carries_the_contract_keys():
    d = _recorded_detail()
    text = bench.compact_line(d)
    assert len(text.encode()) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in bench.CONTRACT_KEYS + ("roofline", "cpu_baseline", "summary", "detail"):
        assert k in line, k
    assert line["metric"] == json.load(open(os.path.join(bench.ROOT, "BASELINE.json")))["metric"]
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms"} <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert abs(line["value"] - d["value"]) <= 1e-4 * d["value"] and line["summary"]["backend"]["cmax_iters_per_s"] > 0
    # the detail's own summary block is regenerated, never copied: nothing of the 21 KB leaks into the line
    assert "kernels" not in line and "per_packet" not in line and "backend" not in line


def test_stdout_line_never_exceeds_the_budget_whatever_the_legs_hold():
    d = _recorded_detail()
    d["config"]["workload"] = "w" * 5000
    d["cpu_baseline"]["sample"] = "s" * 5000
    d["per_packet"] = {"error": "e" * 3000}
    d["backend"]["per_window"] = {"error": "boom " * 400}
    d["error"] = "x" * 2000
    text = bench.compact_line(d)
    assert len(text.encode()) <= bench.LINE_BUDGET
    line = json.loads(text)
    for k in bench.CONTRACT_KEYS:
        assert k in line, k
    assert "roofline" in line and "cpu_baseline" in line and len(line["error"]) <= 300


def test_leg_errors_surface_in_the_line():
    d = _recorded_detail()
    d["per_packet"] = {"error": "RuntimeError('no store')"}
    d["backend"]["launch_defaults"] = {"error": "ValueError('x')"}
    line = json.loads(bench.compact_line(d))
    assert any(e.startswith("per_packet:") for e in line["leg_errors"]) and any("launch_defaults" in e for e in line["leg_errors"])


def test_plan_launch():
    P = bench.plan_launch
    assert P(1, 1, 1) == {"form": "single", "devices": None, "n_gpus": 1, "spawn": 0, "error": None}
    assert P(1, 1, 8)["form"] == "single"
    # torch.distributed.run: the world size wins
    assert P(8, 8, 8)["form"] == "ranks" and P(8, 8, 8)["n_gpus"] == 8 and P(8, 8, 8)["error"] is None
    assert P(4, 2, 8)["n_gpus"] == 2 and "ignored" in P(4, 2, 8)["error"]
    # plain `python bench.py --gpus N`: one process, group handle, process-per-GPU form spawned
    p = P(4, 1, 8)
    assert p["form"] == "group" and p["devices"] == [0, 1, 2, 3] and p["n_gpus"] == 4 and p["spawn"] == 4 and p["error"] is None
    assert P(4, 1, 8, single_process=True)["spawn"] == 0
    # fewer devices than asked for: what is there runs, the line says so
    p = P(2, 1, 1)
    assert p["form"] == "group" and p["devices"] == [0] and p["n_gpus"] == 1 and p["spawn"] == 0 and "2 requested" in p["error"]
    p = P(8, 1, 4)
    assert p["devices"] == [0, 1, 2, 3] and p["n_gpus"] == 4 and "8 requested" in p["error"]
    p = P(2, 1, 0)
    assert p["form"] == "none" and p["n_gpus"] == 0 and p["error"]
    # explicit members (two members sharing device 0 on a one-GPU box)
    p = P(2, 1, 1, group_devices=[0, 0])
    assert p["form"] == "group" and p["devices"] == [0, 0] and p["n_gpus"] == 1 and p["error"] is None
    assert P(2, 1, 1, group_devices=[0, 1])["form"] == "none"


def test_bench_as_the_driver_types_it_ends_in_one_parseable_line_without_a_gpu():
    """`python3 bench.py --gpus 2` (no launcher, here: no device either) must exit 0 with ONE JSON line that says what happened."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for n in ("2", "1"):
        p = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--gpus", n, "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                            "--no-per-packet", "--no-backend", "--solves", "0", "--no-spawn", "--events", "100000", "--detail-out", os.path.join(os.environ.get("TMPDIR", "/tmp"), "bench_detail_test.json")],
                           capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < 4096
        d = json.loads(lines[0])
        for k in bench.CONTRACT_KEYS:
            assert k in d, k
        try:
            import torch
            has_gpu = torch.cuda.is_available()
        except Exception:
            has_gpu = False
        if not has_gpu:
            assert d["n_gpus"] == 0 and "no HIP device" in d["error"]


def test_spawned_process_per_gpu_leg_returns_the_childs_line(tmp_path):
    """The nested leg of a plain `--gpus N`: bench.py re-launched under torch.distributed.run, its ONE stdout line parsed.  On a box
    without a GPU the child still ends in a parseable line (n_gpus 0 + error) -- the plumbing is what is tested here."""
    import argparse
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("CPU-box test of the launcher plumbing (the GPU box runs the real thing in tests/test_gpu_bench_contract.py)")
    except ImportError:
        pass
    a = argparse.Namespace(steps=3, warmup=1, mode="fast", comm="native", events=1000, no_config5=True, no_parity=True,
                           detail_out=str(tmp_path / "bench_detail.json"))
    r = bench.spawn_process_per_gpu(1, a, timeout_s=300)
    assert "value" in r and r["n_gpus"] == 0 and "no HIP device" in r["error"], r
    s = bench.summary_of({"process_per_gpu": r, "ms_per_step": 1.0, "value": 1.0, "n_gpus": 1})
    assert s["process_per_gpu"]["n_gpus"] == 0


def test_stdout_line_of_the_sharded_and_group_forms():
    """The N > 1 shapes of the detail (process-per-GPU dry run with config 5 nested; one-process group with its per-window pipeline):
    the line stays under the budget and carries what a scaling record needs -- comm (nranks_seen), parity vs one GPU, the one-GPU
    reference of the same workload."""
    for name, want in (("r05h_bench_sharded_world1_detail.json", ("comm", "parity_vs_1gpu", "config5", "one_gpu_same_workload_events_per_s")),
                       ("r05h_bench_group00_detail.json", ("comm", "parity_vs_1gpu", "group", "per_window_ratio_to_solve",
                                                           "one_gpu_same_workload_events_per_s"))):
        d = json.load(open(os.path.join(bench.ROOT, "profiles", name)))
        text = bench.compact_line(d)
        assert len(text.encode()) < 4096, name
        line = json.loads(text)
        for k in bench.CONTRACT_KEYS + ("roofline", "summary"):
            assert k in line, (name, k)
        for k in want:
            assert k in line["summary"], (name, k)
        assert line["summary"]["comm"]["nranks_seen"] in (1, 2) and line["summary"]["parity_vs_1gpu"]["grad_rel_inf"] < 1e-5
        assert "config 4" in line["config"]["workload"]
