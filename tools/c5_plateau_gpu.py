"""Where the Deep100M-shape recall plateau occurs, GPU half: the library's batched inserter on the rows of
tools/c5_plateau_cpu.py (same generator, same seeds), searched by the library, against its exact k-NN.
usage: c5_plateau_gpu.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered, recall_at_k
import vectordb_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d, nq, k = 96, 2000, 10
vdb = vectordb_amd.api()
centres = np.random.default_rng(0xC5).uniform(0, 128, size=(max(16, n // 3000), d)).astype(np.float32)
base = clustered(n, d, 1, centres); queries = clustered(nq, d, 2, centres)
ids = np.arange(n, dtype=np.uint64)
print(f"n={n} d={d} centres={len(centres)} ({n // len(centres)} points per centre)", flush=True)
t0 = time.time()
hb = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=7)
for lo in range(0, n, 250_000):
    hb.insert_batch(ids[lo:lo + 250_000], base[lo:lo + 250_000])
print(f"batched GPU build {time.time() - t0:.1f}s", flush=True)
exact, _ = hb.exact_knn_batch(queries, k)
info = hb.device_info()
for width in (0, 1):
    h = hb
    if width:
        h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=7, beam_width=1); h.import_graph(hb.export_graph())
    for ef in (16, 32, 64, 128, 256):
        out, cnt, st = h.search_batch(queries, k, ef_search=ef, with_stats=True)
        print(f"  batched{' (one node per step)' if width else ''} ef={ef:3d} recall@10={recall_at_k(out, cnt, exact, k):.4f} "
              f"rows/query={st['rows_scored'].mean():.0f} deg={info['l0_edges']/n:.1f}", flush=True)
