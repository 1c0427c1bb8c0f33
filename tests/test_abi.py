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


def test_the_device_buffer_scan_refuses_impossible_shapes_before_it_touches_a_device(vdb):
    """vdb_score_scan_dev checks its sizes on the host: a stride shorter than the rows would read past every row."""
    import pytest
    p = 4096                                                   # never dereferenced: the call is refused first
    for nq, n, dim, stride in ((1, 10, 0, 4), (1, 10, 8, 4), (1, 1 << 32, 8, 8), (1 << 32, 10, 8, 8)):
        with pytest.raises(vdb.ArgumentError):
            vdb.check(vdb.fn.score_scan_dev(p, nq, p, n, dim, stride, int(vdb.DistanceMetric.Dot), int(vdb.Arith.Fast), p, None))
