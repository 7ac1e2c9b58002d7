"""-m gpu: the native RCCL communicator with more than one rank (one process per GPU over xGMI).  Skipped on a one-GPU
box -- there the same exchange logic runs with two ranks on one GPU through a host transport
(tests/test_gpu_multirank_sim.py) -- and run automatically by the first box that has two GPUs or more."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    from cmax_slam_amd import _lib
    return _lib.lib().cmx_device_count()


def _launch(nproc, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_worker.py")]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


def test_worker_script_runs_with_one_rank(hip):
    """The worker itself (1-rank communicator): keeps the multi-GPU test from rotting on one-GPU boxes."""
    r = _launch(1, 29541)
    assert r.returncode == 0 and "MULTIRANK_OK world=1" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_native_communicator_multi_gpu(hip, nproc):
    if _device_count() < nproc:
        pytest.skip("needs %d GPUs" % nproc)
    r = _launch(nproc, 29542 + nproc)
    assert r.returncode == 0 and ("MULTIRANK_OK world=%d" % nproc) in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
