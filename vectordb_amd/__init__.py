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
LIB_PATH = os.path.join(_HERE, "csrc", "libvdbmi.so")
_api = None


def load_library():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C vectordb_amd/csrc`; there is no CPU fallback")
    return ctypes.CDLL(LIB_PATH)


def api():
    global _api
    if _api is None:
        _api = capi.bind(load_library(), prefix="vdb_")
    return _api
