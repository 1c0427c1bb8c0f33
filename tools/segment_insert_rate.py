"""Segment::insert one point at a time with a payload (segment.rs:32-60: index insert, payload index, filter-aware
edges of hnsw.rs:210-325) at a given segment size; payload {cat: i mod 100, tier: mix(i) mod 10}."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
extra, d = 1000, 128
x = clustered(n0 + extra, d, 0xC3, np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32))
x = x[np.random.default_rng(2).permutation(len(x))]
def payload(i):
    p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 100)); p.set("tier", vdb.PayloadValue.Int(((i * 0x9E3779B97F4A7C15) >> 40) % 10))
    return p
pl = [payload(i) for i in range(n0 + extra)]
seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=1, filter_mode=vdb.FilterMode.Masked))
t0 = time.time()
for lo in range(0, n0, 100_000): seg.insert_batch(x[lo:min(n0, lo + 100_000)], pl[lo:min(n0, lo + 100_000)])
print(f"batch build of {n0} points with payloads: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
ids = [seg.insert(x[i], pl[i]) for i in range(n0, n0 + extra)]
dt = time.time() - t0
f = vdb.Filter.Match("cat", vdb.PayloadValue.Int((n0 + 5) % 100))
out = seg.search_with_filter(x[n0 + 5], 1, f)
assert out and out[0].id == ids[5], "a freshly inserted point is found under its own filter"
print(f"{extra/dt:.0f} single segment inserts/s at {n0} points ({dt/extra*1e3:.2f} ms each)")
