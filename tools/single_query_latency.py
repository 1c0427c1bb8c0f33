"""Latency of the reference's own call shape: one query per call through HNSWIndex::search (host buffers)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(2000, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=0xC2)      # ef = 64, the index default
h.insert_batch(np.arange(n, dtype=np.uint64), base)
for i in range(50): h.search(q[i], 10)
t0 = time.perf_counter()
for i in range(2000): h.search(q[i], 10)
dt = (time.perf_counter() - t0) / 2000
print(f"one query per call, {n} x {d}, ef 64, top-10 (over-fetch 20): {dt*1e6:.0f} us per call = {1/dt:.0f} calls/s")
