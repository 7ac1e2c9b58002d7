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
    import types

    # The fixture hands out the LIBRARY'S OWN DEFAULT: the production path (adjoint gradient, LDS-privatised splat).  A test that
    # forgets to choose a path therefore tests the production kernels.  The reference-shaped path (derivative planes, one global
    # atomic per vote: the reference's data flow, and the second GPU implementation the production path is checked against) is
    # an explicit opt-in, visible in the test's text: hip.reference_shaped.FrontendEvaluator / .BackendEvaluator.
    class RefFrontendEvaluator(evaluator.FrontendEvaluator):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.set_reference_path()

    class RefBackendEvaluator(evaluator.BackendEvaluator):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.set_reference_path()

    ns = types.SimpleNamespace(**{k: v for k, v in vars(evaluator).items() if not k.startswith("__")})
    ns.reference_shaped = types.SimpleNamespace(FrontendEvaluator=RefFrontendEvaluator, BackendEvaluator=RefBackendEvaluator)
    return ns
