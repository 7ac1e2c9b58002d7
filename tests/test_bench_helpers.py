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
