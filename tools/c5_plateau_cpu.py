"""Where the Deep100M-shape recall plateau occurs, CPU half: the oracle's point-by-point build (hnsw.rs:119-208
literally) of n x 96 L2 rows at 3000 points per Gaussian centre, searched by the oracle at several ef, against a float64
re-ranked brute force.  tools/c5_plateau_gpu.py builds the same rows with the library's batched inserter and prints the
same table; both logs go to profiles/.  Test tooling: runs on host cores only.
usage: c5_plateau_cpu.py [n] [threads]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered, recall_at_k
from oracle import pyoracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, nq, k = 96, 2000, 10
ora = pyoracle.load()
centres = np.random.default_rng(0xC5).uniform(0, 128, size=(max(16, n // 3000), d)).astype(np.float32)
base = clustered(n, d, 1, centres); queries = clustered(nq, d, 2, centres)
ids = np.arange(n, dtype=np.uint64)
print(f"n={n} d={d} centres={len(centres)} ({n // len(centres)} points per centre)", flush=True)

# ground truth: float32 expansion picks 64 candidates per query, float64 re-ranks them
t0 = time.time()
bn = (base.astype(np.float64) ** 2).sum(1)
cand = np.empty((nq, 64), np.int64); cd = np.full((nq, 64), np.inf)
for lo in range(0, n, 500_000):
    hi = min(n, lo + 500_000)
    dd = bn[None, lo:hi] - 2.0 * (queries @ base[lo:hi].T).astype(np.float64)
    part = np.argpartition(dd, 63, axis=1)[:, :64]
    alld = np.concatenate([cd, np.take_along_axis(dd, part, 1)], 1); alli = np.concatenate([cand, part + lo], 1)
    sel = np.argpartition(alld, 63, axis=1)[:, :64]
    cd = np.take_along_axis(alld, sel, 1); cand = np.take_along_axis(alli, sel, 1)
exact = np.zeros((nq, k), dtype=[("id", "<u8"), ("raw_score", "<f4"), ("sort_key", "<f4")])
for i in range(nq):
    dist = np.sqrt(((base[cand[i]].astype(np.float64) - queries[i].astype(np.float64)) ** 2).sum(1))
    o = np.lexsort((cand[i], dist))[:k]
    exact["id"][i] = cand[i][o]; exact["sort_key"][i] = dist[o]; exact["raw_score"][i] = dist[o]
print(f"ground truth in {time.time() - t0:.0f}s", flush=True)

t0 = time.time()
ho = ora.HNSWIndex(ora.DistanceMetric.Euclidean, 16, 64, 16, d, seed=7)
for lo in range(0, n, 100_000):
    ho.insert_batch(ids[lo:lo + 100_000], base[lo:lo + 100_000])
    print(f"  oracle inserted {min(n, lo + 100_000)} ({time.time() - t0:.0f}s)", flush=True)
print(f"sequential oracle build {time.time() - t0:.0f}s", flush=True)
ora.set_threads(threads)
for ef in (16, 32, 64, 128, 256):
    out, cnt, st = ho.search_batch(queries, k, ef_search=ef, with_stats=True)
    print(f"  sequential ef={ef:3d} recall@10={recall_at_k(out, cnt, exact, k):.4f} rows/query={st['rows_scored'].mean():.0f}", flush=True)
