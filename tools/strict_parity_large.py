"""Parity at benchmark size: the graph the GPU builds for the 1M x 128 workload is handed to the CPU oracle, and the
STRICT search of the library (and the beam_width = 1 FAST search) is compared with the oracle's search query by
query.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
from oracle import pyoracle
vdb = vectordb_amd.api(); ora = pyoracle.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 144
d, nq, k = 128, 10000, 10
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(nq, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=0xC2)
h.insert_batch(np.arange(n, dtype=np.uint64), base)
g = h.export_graph()
ho = ora.HNSWIndex(ora.DistanceMetric.Euclidean, 16, 64, 16, d); ho.import_graph(g)
ora.set_threads(min(os.cpu_count() or 1, 16))
ref, refn, rst = ho.search_batch(q, k, ef_search=ef, with_stats=True)
res = {"workload": f"{nq} queries x {n} x {d} L2, ef {ef}, top-{k}, graph built by the library"}
for name, kw in (("strict", dict(arith=vdb.Arith.Strict)), ("fast_width1", dict(arith=vdb.Arith.Fast, beam_width=1)), ("fast_default", dict(arith=vdb.Arith.Fast))):
    p = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, **kw); p.import_graph(g)
    p.search_batch(q[:64], k, ef_search=ef)
    t0 = time.perf_counter(); got, gotn, st = p.search_batch(q, k, ef_search=ef, with_stats=True); sec = time.perf_counter() - t0
    res[name] = {"seconds_host_buffers": round(sec, 4), "counts_equal": bool(np.array_equal(refn, gotn)),
                 "id_positions_equal": float(np.mean(ref["id"] == got["id"])),
                 "queries_fully_equal": float(np.mean((ref["id"] == got["id"]).all(axis=1))),
                 "sort_key_bits_equal": float(np.mean(ref["sort_key"].view(np.uint32) == got["sort_key"].view(np.uint32))),
                 "expansions_equal": float(np.mean(rst["expanded"] == st["expanded"])),
                 "max_rel_key_error": float(np.max(np.abs(ref["sort_key"].astype(np.float64) - got["sort_key"]) / np.maximum(ref["sort_key"], 1e-6)))}
print(json.dumps(res))
