"""ORACLE — TEST INFRASTRUCTURE ONLY.

Loads the CPU restatement of the reference search path (oracle/ora_core.hpp) and exposes it
through the same Python classes as the product library.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvdb_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ora_core.hpp", "ora_capi.cpp", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def load():
    sys.path.insert(0, os.path.dirname(_HERE))
    from vectordb_amd import capi
    # VDB_ORACLE_LIBRARY: another build of the same sources (tools/sanitize_host.sh: AddressSanitizer + UBSan)
    lib = ctypes.CDLL(os.environ.get("VDB_ORACLE_LIBRARY") or build())
    api = capi.bind(lib, prefix="ora_", required=set())
    api.set_threads = api.fn.set_threads
    return api
