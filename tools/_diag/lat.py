import os, sys, time, numpy as np
sys.path.insert(0, '.')
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
n, d = 1_000_000, 128
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(2000, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=0xC2)
h.insert_batch(np.arange(n, dtype=np.uint64), base)
def lat(**env):
    for k, v in env.items(): os.environ[k] = str(v)
    for i in range(50): h.search(q[i], 10)
    t0 = time.perf_counter()
    for i in range(2000): h.search(q[i], 10)
    dt = (time.perf_counter() - t0) / 2000
    for k in env: del os.environ[k]
    return dt * 1e6
print("team 4 zero-copy", lat()); print("team 4 copies", lat(VDB_NO_ZERO_COPY=1)); print("single zero-copy", lat(VDB_TEAM=0)); print("single copies", lat(VDB_TEAM=0, VDB_NO_ZERO_COPY=1))
print("team 4 width 8", lat(VDB_TEAM_WIDTH=8)); print("team 2", lat(VDB_TEAM=2))
a = [h.search(q[i], 10) for i in range(100)]
os.environ["VDB_NO_ZERO_COPY"] = "1"; b = [h.search(q[i], 10) for i in range(100)]
print("same", all([r.id for r in x] == [r.id for r in y] for x, y in zip(a, b)))
