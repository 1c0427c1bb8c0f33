"""Quality gate for the batched GPU graph builder: recall@10 of a graph built by the library's
insert_batch (rounds search a frozen graph) against a graph built point by point by the CPU oracle
(the reference's insert, hnsw.rs:119-208), same data, same M / ef_construction.  Test tooling."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered, recall_at_k
import vectordb_amd
from oracle import pyoracle

# usage: build_quality.py [n] [dim] [number of Gaussian centres]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nq, k = 2000, 10
vdb, ora = vectordb_amd.api(), pyoracle.load()
n_centres = int(sys.argv[3]) if len(sys.argv) > 3 else max(16, n * 4096 // 1_000_000)
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(n_centres, d)).astype(np.float32)
base = clustered(n, d, 1, centres); queries = clustered(nq, d, 2, centres)
ids = np.arange(n, dtype=np.uint64)
M = vdb.DistanceMetric.Euclidean

t0 = time.time()
hb = vdb.HNSWIndex(M, 16, 64, 16, d, seed=7)
hb.insert_batch(ids, base)
tb = time.time() - t0
t0 = time.time()
ho = ora.HNSWIndex(M, 16, 64, 16, d, seed=7)
for lo in range(0, n, 50_000):                      # in pieces, with a progress line: a long silent run looks hung
    ho.insert_batch(ids[lo:lo + 50_000], base[lo:lo + 50_000])
    print(f"  oracle inserted {min(n, lo + 50_000)} ({time.time() - t0:.0f}s)", flush=True)
to = time.time() - t0
hs = vdb.HNSWIndex(M, 16, 64, 16, d, seed=7)
hs.import_graph(ho.export_graph())
exact, _ = hb.exact_knn_batch(queries, k)
print(f"n={n} d={d} centres={n_centres} ({n // n_centres} points per centre): batched GPU build {tb:.1f}s, sequential oracle build {to:.1f}s")
for name, h in (("batched", hb), ("sequential", hs)):
    info = h.device_info()
    for ef in (16, 32, 64, 128, 256):
        out, cnt, st = h.search_batch(queries, k, ef_search=ef, with_stats=True)
        print(f"  {name:10s} ef={ef:3d} recall@10={recall_at_k(out, cnt, exact, k):.4f} rows/query={st['rows_scored'].mean():.0f} "
              f"p99={np.percentile(st['rows_scored'], 99):.0f} max={st['rows_scored'].max()} deg={info['l0_edges']/n:.1f}")
