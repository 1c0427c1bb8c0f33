"""The C-ABI library loads and exports every symbol include/vdbmi.h declares (no compute calls: CPU-safe)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vdbmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    import ctypes
    import vectordb_amd
    lib = vectordb_amd.load_library()
    names = declared_symbols()
    assert len(names) > 80
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binder_covers_header():
    from vectordb_amd import capi
    bound = {"vdb_" + n for n in capi.ABI_SYMBOLS}
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_no_gpu_means_loud_failure(vdb):
    """Without a GPU the product must fail with a device error, never fall back to CPU arithmetic."""
    import pytest
    if vdb.fn.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vdb.DeviceError):
        vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, 4)
    with pytest.raises(vdb.DeviceError):
        vdb.score([1.0, 2.0], [3.0, 4.0], vdb.DistanceMetric.Dot)
