"""Single-point insert rate (HNSWIndex::insert one by one, the reference's usage) at a given index size.
Second argument: `clustered` (the benchmark's generator, default) or `uniform` (uniform(0,1) coordinates: hub lists
with thousands of entries, the hard case for a walk)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectordb_amd
vdb = vectordb_amd.api()
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = 128
kind = sys.argv[2] if len(sys.argv) > 2 else "clustered"
if kind == "uniform":
    x = np.random.default_rng(1).uniform(0, 1, size=(n0 + 2000, d)).astype(np.float32)
else:
    from bench import clustered
    x = clustered(n0 + 2000, d, 0xC3, np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32))
    x = x[np.random.default_rng(2).permutation(len(x))]
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d)
h.insert_batch(np.arange(n0, dtype=np.uint64), x[:n0])
t0 = time.time()
for i in range(n0, n0 + 2000):
    h.insert(i, x[i])
dt = time.time() - t0
t0 = time.time()
for i in range(n0 + 1000, n0 + 2000):          # mixed traffic: every insert is followed by a search
    h.search(x[i], 10)
    h.insert(i + 1000000000, x[i] * np.float32(0.999))
dm = time.time() - t0
q = x[n0:n0 + 100]
out, cnt = h.search_batch(q, 1)
assert (out["id"][:, 0] == np.arange(n0, n0 + 100)).mean() > 0.95, "freshly inserted points must be found"
print(f"alternating insert + search: {dm/1000*1e3:.2f} ms per pair")
print(f"{2000/dt:.0f} single inserts/s at {n0} {kind} points ({dt/2000*1e3:.2f} ms each)")
