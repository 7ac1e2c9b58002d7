"""profiles/pmc_traffic.json from a tools/gpu_profile.sh summary: per-launch memory-side bytes of the bench's kernels.

FETCH_SIZE / WRITE_SIZE are in KiB (checked: bearing_stream_kernel writes 5M x 16 B = 80.0 MB, WRITE_SIZE = 78125.0).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read -- confirmed here on fe_splat_lds (streams 24 B/event = 24.0 MB, FETCH_SIZE = 12.2 MB) and be_gather4
(100 MB of 16-B/lane streams, FETCH_SIZE = 53.3 MB).  `bytes` = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the kernels whose
reads are such streams; `raw_bytes` = (FETCH_SIZE + WRITE_SIZE) * 1024 is kept beside it; kernels with mixed access widths
(be_splat_lds: 4-byte + 16-byte lanes + divergent rotation gathers) list both and bench.py reports the corrected one.
    python tools/pmc_to_json.py gpurun_out/prof_<tag>/summary.txt"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# FETCH_SIZE under-reports by a factor that depends on the access width (tools/microbench/fetch_calib.hip under
# rocprofv3 --pmc FETCH_SIZE, profiles/r03_fetch_calib.txt): known bytes / reported bytes per stream mix.  Applied per
# kernel below instead of a blanket x2.
FETCH_FACTOR = {"frontend_fast_splat": 2.0, "frontend_fast_gather": 2.0, "backend_fast_gather": 2.0, "backend_fast_splat": 2.0,
                "backend_fast_pose": 2.0, "backend_fast_batch": 2.0}
try:
    FETCH_FACTOR.update(json.load(open(os.path.join(ROOT, "profiles", "fetch_calibration.json")))["factor_by_kernel"])
except Exception:
    pass


def source_hash():
    """sha256 over the product's kernel / host sources: what bench.py recomputes to decide whether this file describes the build
    it is running (the GPU box has no .git)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cmax_slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def so_hash():
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "cmax_slam_amd", "libcmaxhip.so"), "rb").read()).hexdigest()
    except OSError:
        return None

# coalesced per-event STREAM bytes one launch of each kernel reads at the bench's sizes (1M front-end / 5M back-end events): the
# part of FETCH_SIZE that is known to be reported at half (profiles/r03_fetch_calib.txt).  What is left of FETCH_SIZE after it is
# table-gather traffic (L2 misses of divergent 8-byte loads) whose request size -- hence factor, 1 or 2 -- is not known:
# `bytes_lower` counts it once, `bytes` twice; for the stream-only kernels the two coincide.
STREAM_BYTES = {"frontend_fast_splat": 24e6, "frontend_fast_gather": 24e6, "backend_fast_splat": 120e6, "backend_fast_gather": 100e6}

NAMES = {
    "fe_splat_lds_kernel<false, true, 1>": "frontend_fast_splat",  # round 6: the splat launch that carries the image pass (the timed fdf evaluations);
                                                                     # <.., 0> is the plain splat of cost-only evaluations
    "fe_gather_kernel<0>": "frontend_fast_gather",  # the evaluations bench.py times (<1>, <2>: the device-driven solves, incl. gated-off launches)
    "image_adjoint_kernel<4, 64, 16, 1024, false>": None,  # shared by both ends in one run: split by call order is not possible
    "be_splat_lds_kernel": "backend_fast_splat", "be_gather4_kernel": "backend_fast_gather",
    "be_gather_batch_kernel": "backend_fast_batch", "be_pose_table_pre_kernel<4, true>": "backend_fast_pose",
}


def main(path):
    vals = {}
    for line in open(path):
        m = re.match(r"(.{58})\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU_[A-Z0-9_]+|SQ_INSTS_VALU|SQ_INSTS_LDS|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_BUSY_CYCLES|SQ_WAVES|SQ_WAVE_CYCLES)\s+([\d.]+)", line)
        if not m:
            continue
        vals.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "cmax_slam_amd/csrc"]).strip())
    except Exception:
        head, dirty = None, None
    src, so = source_hash(), so_hash()
    try:  # the hashes taken ON THE GPU BOX when the counters were collected (tools/gpu_profile.sh)
        src, so = open(os.path.join(os.path.dirname(path), "stamp.txt")).read().split()[:2]
    except Exception:
        pass
    if src != source_hash():
        print("WARNING: the sources changed since the counters were collected; git_head below is only informational", file=sys.stderr)
    out = {"_note": __doc__.split("\n    python")[0], "_source": os.path.basename(os.path.dirname(path)) + "/summary.txt",
           "_stamp": {"src_sha256": src, "so_sha256": so, "git_head": head, "csrc_dirty_vs_head": dirty,
                      "note": "bench.py reports these byte counts only when its own source hash of cmax_slam_amd/csrc equals src_sha256"},
           "kernels": {}}
    for kname, d in vals.items():
        for pat, key in NAMES.items():
            if key and pat in kname and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                raw = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
                fac = FETCH_FACTOR.get(key, 2.0)
                cor = (fac * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
                row = {"fetch_kib": d["FETCH_SIZE"], "write_kib": d["WRITE_SIZE"], "raw_bytes": raw, "fetch_factor": fac, "bytes": cor}
                if key in STREAM_BYTES:
                    fetch = d["FETCH_SIZE"] * 1024
                    stream_raw = min(STREAM_BYTES[key] / 2, fetch)
                    row["stream_bytes_known"] = STREAM_BYTES[key]
                    row["bytes_lower"] = 2 * stream_raw + (fetch - stream_raw) + d["WRITE_SIZE"] * 1024
                # instruction issue of the same launches (separate PMC passes): wave-level VALU instructions, and the clocks the VALU
                # pipes were issuing (SQ_ACTIVE_INST_VALU counts in units of 4 clocks, summed over the chip's SIMDs)
                for cname, oname in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_ACTIVE_INST_VALU", "valu_active_x4clk"),
                                     ("SQ_ACTIVE_INST_LDS", "lds_active_x4clk"), ("SQ_BUSY_CYCLES", "sq_busy_cycles"), ("SQ_WAVES", "waves"),
                                     ("SQ_WAVE_CYCLES", "wave_cycles_x4clk")):
                    if cname in d:
                        row[oname] = d[cname]
                mix = {c[len("SQ_INSTS_VALU_"):].lower(): v for c, v in d.items() if c.startswith("SQ_INSTS_VALU_")}
                if mix:  # wave-level instructions per class (separate PMC passes of the same command)
                    row["valu_mix"] = mix
                    out[key + "_valu_mix"] = mix
                out["kernels"][key] = row
                out[key] = cor
                if "valu_insts" in row:
                    out[key + "_valu_insts"] = row["valu_insts"]
                if "valu_active_x4clk" in row:
                    out[key + "_valu_active_x4clk"] = row["valu_active_x4clk"]
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
