"""CPU: the C-ABI shared library loads and exports every symbol include/cmax_hip.h declares; without a GPU the
product path fails loudly (there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from cmax_slam_amd import _lib
    _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    from cmax_slam_amd import _lib
    # the host surface + the diagnostic header (A/B option keys, the two static scheduling calls, process-wide test switches)
    hdr = open(os.path.join(ROOT, "include", "cmax_hip.h")).read() + open(os.path.join(ROOT, "include", "cmax_hip_diag.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cmx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(raw, name), "libcmaxhip.so does not export %s" % name
    assert declared == set(_lib.SYMBOLS), "python binding table and header disagree: %s" % (declared ^ set(_lib.SYMBOLS))


def test_version_and_status_strings(L):
    assert b"gfx950" in L.cmx_version()
    assert L.cmx_status_string(0) == b"ok"
    for s in range(1, 7):
        assert len(L.cmx_status_string(s)) > 3


def test_traj_temp_start_truncates_like_the_reference(L, oracle):
    # int64_t(1e9 * (t_beg + idx*dt)): (double)->ns truncation (trajectory.cpp:255-256)
    for t_beg, idx, dt in ((1.0, 3, 0.05), (1700000000.123456789, 7, 0.05), (12.3456, 0, 0.01)):
        assert L.cmx_traj_temp_start_ns(t_beg, idx, dt) == int(1e9 * (t_beg + idx * dt))
        assert L.cmx_traj_temp_start_ns(t_beg, idx, dt) == oracle.traj_temp_start_ns(t_beg, idx, dt)


def test_no_gpu_means_loud_failure_not_fallback(L):
    if L.cmx_device_count() > 0:
        pytest.skip("a GPU is visible; the failure path is covered on CPU-only boxes")
    from cmax_slam_amd import evaluator
    with pytest.raises(evaluator.CmaxHipError) as e:
        evaluator.FrontendEvaluator(8, 8, np.zeros(8 * 8 * 3))
    assert e.value.status == 3
    with pytest.raises(evaluator.CmaxHipError):
        evaluator.BackendEvaluator(8, 8, np.zeros(8 * 8 * 3), 64, 32)


def test_product_package_never_imports_the_oracle():
    """The oracle is the checker: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it."""
    for sub in ("cmax_slam_amd", "examples", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "pyoracle" not in src and "liboracle" not in src and "cmax_oracle" not in src, (sub, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # one import, inside cpu_baseline()
    assert bench.count("pyoracle") == 1 and bench.split("pyoracle")[0].rsplit("\ndef ", 1)[-1].startswith("cpu_baseline(")
