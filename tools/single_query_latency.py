"""Latency of the reference's own call shape: one query per call through HNSWIndex::search (host buffers).
Two figures: the Python binding's call (numpy conversion, result objects) and the C entry point called with prepared
ctypes buffers (what a Rust or C caller sees)."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
from vectordb_amd.capi import ScoredPointC
vdb = vectordb_amd.api()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(2000, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=0xC2)      # ef = 64, the index default
h.insert_batch(np.arange(n, dtype=np.uint64), base)
buf = (ScoredPointC * 10)(); cnt = C.c_size_t(0)
ptrs = [q[i].ctypes.data_as(C.c_void_p) for i in range(2000)]


def measure():
    for i in range(50): h.search(q[i], 10)
    t0 = time.perf_counter()
    for i in range(2000): h.search(q[i], 10)
    py = (time.perf_counter() - t0) / 2000
    t0 = time.perf_counter()
    for i in range(2000): vdb.fn.index_search(h._h, ptrs[i], d, 10, C.cast(buf, C.c_void_p), C.byref(cnt))
    c = (time.perf_counter() - t0) / 2000
    return py * 1e6, c * 1e6


for name, env in (("team of 4 wavefronts (default)", {}), ("one wavefront per query", {"VDB_TEAM": "0"}), ("one wavefront, staged copies", {"VDB_TEAM": "0", "VDB_NO_ZERO_COPY": "1"})):
    os.environ.update(env)
    py, c = measure()
    for k in env: del os.environ[k]
    print(f"one query per call, {n} x {d}, ef 64, top-10, {name}: {py:.0f} us per call through the Python binding, {c:.0f} us through the C entry point")
