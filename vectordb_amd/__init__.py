"""vectordb_amd — MI355X-native search path behind the reference's segment API.

`vectordb_amd.api()` loads libvdbmi.so (hand-written HIP for gfx950, built in-tree by
`__graft_entry__.build()` / `make -C vectordb_amd/csrc`) and returns the API namespace.  There is
no CPU implementation: importing the API without the built library raises ImportError.
"""
import ctypes
import os

from . import capi
from .capi import (Arith, DBError, DistanceMetric, FilterMode, ListQueryKind,  # noqa: F401
                   ScalarComparisonOp, SCORED_DTYPE, STATS_DTYPE)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDB_LIBRARY") or os.path.join(_HERE, "csrc", "libvdbmi.so")   # VDB_LIBRARY: diagnostic builds
_api = None


def _share_torch_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64 (same
    SONAME as the system one); if libvdbmi pulled in the system copy first, a later `import torch`
    would bring a second runtime that finds no GPU.  Loading torch's copy first (without importing
    torch) makes both resolve to it."""
    import importlib.util
    if "torch" in __import__("sys").modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load_library():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C vectordb_amd/csrc`; there is no CPU fallback")
    _share_torch_hip_runtime()
    return ctypes.CDLL(LIB_PATH)


def api():
    global _api
    if _api is None:
        _api = capi.bind(load_library(), prefix="vdb_")
    return _api
