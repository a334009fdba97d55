"""dgll_amd -- MI355X-native sparse GNN aggregation engine behind the dgll.nn conv-layer API.

Importing this package loads libdgll_hip.so (hand-written HIP for gfx950) through ctypes and fails loudly
if it is missing.  Host-side mirror of the reference interface lives in dgll_amd.backend (the `F` object of
/root/reference/dgll/__init__.py:1), dgll_amd.nn (conv layers), dgll_amd.data / sampling / dataloader.
"""
from . import _lib  # noqa: F401  (must succeed: no CPU fallback for CUDA tensors)
from .graph import CSRGraph  # noqa: F401
from . import ops  # noqa: F401
from . import backend  # noqa: F401

__version__ = "0.1.0"
