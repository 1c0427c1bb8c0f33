"""Diagnostic: what the filter-aware edges of payload-carrying points (hnsw.rs:210-325) do to degree and search cost."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
n, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000, 128, 10000, 10
centres = np.random.default_rng(0xC4).uniform(0, 128, size=(max(16, n * 4096 // 1_000_000), d)).astype(np.float32)
base = clustered(n, d, 0xC4 + 1, centres); q = clustered(nq, d, 0xC4 + 2, centres)
ids = np.arange(n, dtype=np.uint64)
plain = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=1)
plain.insert_batch(ids, base)
pay = []
for i in range(n):
    p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 100)); p.set("tier", vdb.PayloadValue.Int((i * 2654435761 >> 7) % 10)); pay.append(p)
seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=1))
for lo in range(0, n, 100_000): seg.insert_batch(base[lo:lo + 100_000], pay[lo:lo + 100_000])
hs = seg.hnsw()
exact, _ = plain.exact_knn_batch(q, k)
def recall(out, cnt):
    return sum(len(set(out["id"][i][:cnt[i]].tolist()) & set(exact["id"][i].tolist())) for i in range(nq)) / (nq * k)
for name, h, shift in (("plain index", plain, 0), ("segment with 2 payload keys", hs, 1)):
    info = h.device_info()
    g = h.export_graph(); ow, of, nb = g["layers"][0]; deg = np.diff(of.astype(np.int64))
    for ef in (64, 128):
        h.search_batch(q, k, ef_search=ef)
        t0 = time.perf_counter(); out, cnt, st = h.search_batch(q, k, ef_search=ef, with_stats=True); dt = time.perf_counter() - t0
        out = out.copy(); out["id"] -= shift
        print(f"{name}: packed degree {info['l0_edges']/n:.1f} (raw lists: mean {deg.mean():.1f}, above 62: {100*np.mean(deg > 63):.1f} %), ef {ef}: "
              f"{dt*1e3:.2f} ms per 10k queries (host buffers), recall {recall(out, cnt):.4f}, rows/query {st['rows_scored'].mean():.0f}, expansions {st['expanded'].mean():.0f}")
