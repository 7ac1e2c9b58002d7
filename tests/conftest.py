import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The HIP evaluator package; fails loudly if libcmaxhip.so is missing or no GPU is visible."""
    from cmax_slam_amd import _lib
    L = _lib.lib()
    assert L.cmx_device_count() > 0, "no HIP device visible: -m gpu tests need a GPU"
    from cmax_slam_amd import evaluator
    return evaluator
