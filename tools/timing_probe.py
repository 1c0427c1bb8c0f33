"""Diagnostic (needs the -DVDB_TIMING build of the library): per-query wavefront lifetime vs work."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
# usage: timing_probe.py [rows] [ef] [dim] [queries] [centres] [metric]   (defaults: the 1M x 128 L2 benchmark; C3: 1000000 384 960 1000 1024 0)
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 10
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 192
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
nc = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
metric = int(sys.argv[6]) if len(sys.argv) > 6 else 2
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(nc, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(nq, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric(metric), 16, 64, 16, d, seed=0xC2)
for s0 in range(0, n, 250_000):
    h.insert_batch(np.arange(s0, min(n, s0 + 250_000), dtype=np.uint64), base[s0:s0 + 250_000])
for _ in range(2):
    t0 = time.perf_counter()
    out, cnt, st = h.search_batch(q, k, ef_search=ef, with_stats=True)
    wall = time.perf_counter() - t0
life = st["adj_entries"].astype(np.float64) * 0.01      # us
start = st["rows_scored"].astype(np.int64); start = (start - start.min()) * 0.01
print(f"host wall {wall*1e3:.2f} ms; kernel span {(start + life).max()/1e3:.3f} ms")
print(f"wave lifetime us: mean {life.mean():.0f} p50 {np.median(life):.0f} p90 {np.percentile(life,90):.0f} p99 {np.percentile(life,99):.0f} max {life.max():.0f}")
print(f"expansions: mean {st['expanded'].mean():.1f} p99 {np.percentile(st['expanded'],99):.0f} max {st['expanded'].max()}")
print(f"us per expansion: mean {(life/np.maximum(st['expanded'],1)).mean():.2f}")
print(f"sum of lifetimes / span = mean resident waves {life.sum()/(start+life).max():.0f}")
order = np.argsort(start)
print("start-time deciles (us):", np.percentile(start, [0,10,50,90,99,100]).round(0))
ph = out.view(np.uint8).reshape(nq, -1)[:, :80].copy().view(np.uint64).astype(np.float64)
names = ["select", "adjacency wait", "compaction", "score(gather)", "merge", "-", "bucket read+classify", "CAS", "resolve+slow path", "-"]
print(f"before the beam (query load, upper-layer descent): mean {ph[:, 9].mean()*0.01:.1f} us of {ph[:, 5].mean()*0.01:.1f} us per query")
if os.environ.get("VDB_TIMING_SCORE"):
    print(f"team, wavefront 0: candidates per step-round {ph[:, 8].sum() / np.maximum(st['expanded'].sum() / 4, 1):.1f} (per 4 expansions); row wait {ph[:, 6].sum()/st['expanded'].sum():.0f} cycles/expansion, scoring arithmetic {ph[:, 7].sum()/st['expanded'].sum():.0f}")
    ph[:, 3] -= ph[:, 6] + ph[:, 7]; ph[:, 8] = 0
ph[:, 9] = 0; ph[:, 5] = 0
tot = ph.sum(axis=1)
print("phase shares of beam cycles (mean over queries):")
for i, nm in enumerate(names):
    print(f"  {nm:14s} {100*ph[:, i].sum()/tot.sum():5.1f}%   cycles/expansion {ph[:, i].sum()/st['expanded'].sum():7.0f}")
print(f"  beam cycles per expansion {tot.sum()/st['expanded'].sum():.0f}; implied clock {tot.mean()/life.mean():.0f} MHz-equivalent (cycles per us of lifetime)")
rs = st_rows = None
out2, cnt2, st2 = h.search_batch(q[:1], k, ef_search=ef, with_stats=True)
big = np.argsort(-life)[:8]
print("slowest queries: lifetime us / expansions:", [(int(life[i]), int(st['expanded'][i])) for i in big])
