"""Single-point insert rate (Segment::insert one by one, the reference's usage) at a given index size."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectordb_amd
vdb = vectordb_amd.api()
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = 128
rng = np.random.default_rng(1)
x = rng.uniform(0, 1, size=(n0 + 2000, d)).astype(np.float32)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d)
h.insert_batch(np.arange(n0, dtype=np.uint64), x[:n0])
t0 = time.time()
for i in range(n0, n0 + 2000):
    h.insert(i, x[i])
dt = time.time() - t0
q = x[n0:n0 + 100]
out, cnt = h.search_batch(q, 1)
assert (out["id"][:, 0] == np.arange(n0, n0 + 100)).mean() > 0.95, "freshly inserted points must be found"
print(f"{2000/dt:.0f} single inserts/s at {n0} points ({dt/2000*1e3:.2f} ms each)")
