"""cmax_slam_amd -- MI355X-native evaluator for cmax_slam's event-warping hot path.

The package holds only what the path needs: csrc/ (HIP kernels + the C ABI of include/cmax_hip.h),
the host-side mirror of the reference's evaluator interface (evaluator.py), the FR-CG driver the
reference runs around it (solver.py), event sharding across GPUs (dist.py) and synthetic data (synth.py).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
