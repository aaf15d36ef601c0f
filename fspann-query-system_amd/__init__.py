"""fspann-query-system_amd — MI355X-native TokenGen -> Route -> Refine for FSPANN.

The directory name is not a valid Python identifier; load it with
`__graft_entry__.load_package()` (registers it as module `fspann_amd`).
Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI),
_native.py (ctypes), engine.py (numpy batch API), operators.py (mirror of the
reference's Java operator surface), dist.py (query sharding + RCCL top-k merge).
"""
from . import _native  # noqa: F401
from ._native import (FspannArgumentError, FspannDeviceError, FspannError, FspannNullError,  # noqa: F401
                      FspannRangeError, FspannStateError, build)
from .engine import FspannContext, PaperRuntimeConfig  # noqa: F401
