import os, sys, time, numpy as np
sys.path.insert(0, '.')
from bench import clustered, recall_at_k
import vectordb_amd
vdb = vectordb_amd.api()
def run(h, q, k, ef, **env):
    for kk, v in env.items(): os.environ[kk] = str(v)
    out = h.search_batch(q, k, ef_search=ef, with_stats=True)
    t0 = time.perf_counter(); out = h.search_batch(q, k, ef_search=ef, with_stats=True); dt = time.perf_counter() - t0
    for kk in env: del os.environ[kk]
    return out, dt
for (n, d, metric, nq, ef) in ((200_000, 128, 2, 1000, 144), (100_000, 960, 0, 500, 384), (100_000, 96, 1, 300, 64), (50_000, 1536, 2, 200, 100)):
    centres = np.random.default_rng(1).uniform(0, 128, size=(max(16, n // 250), d)).astype(np.float32)
    base = clustered(n, d, 2, centres); q = clustered(nq, d, 3, centres)
    h = vdb.HNSWIndex(metric, 16, 64, 16, d, seed=5)
    h.insert_batch(np.arange(n, dtype=np.uint64), base)
    exact, _ = h.exact_knn_batch(q, 10)
    (ref, refn, rst), t_ref = run(h, q, 10, ef, VDB_TEAM=0)
    print(f"n={n} d={d} metric={metric} nq={nq} ef={ef}: single {t_ref*1e3:.2f} ms recall {recall_at_k(ref, refn, exact, 10):.4f} rows {rst['rows_scored'].mean():.0f} exp {rst['expanded'].mean():.0f}")
    for tw in (2, 4):
        (g, gn, gst), t = run(h, q, 10, ef, VDB_TEAM=tw, VDB_TEAM_WIDTH=4)
        same = np.mean(g["id"] == ref["id"]); keq = np.array_equal(g["sort_key"].view(np.uint32), ref["sort_key"].view(np.uint32))
        print(f"   team {tw} width 4: {t*1e3:.2f} ms ids equal {same:.5f} keys bitwise {keq} counts {np.array_equal(gn, refn)} rows {gst['rows_scored'].mean():.0f} exp {gst['expanded'].mean():.0f} (exp equal {np.mean(gst['expanded']==rst['expanded']):.4f})")
        (g, gn, gst), t = run(h, q, 10, ef, VDB_TEAM=tw)
        print(f"   team {tw} width {4*tw}: {t*1e3:.2f} ms recall {recall_at_k(g, gn, exact, 10):.4f} rows {gst['rows_scored'].mean():.0f} exp {gst['expanded'].mean():.0f}")
    (g1, g1n, _), t1 = run(h, q[:1], 10, ef, VDB_TEAM=0); (g4, g4n, _), t4 = run(h, q[:1], 10, ef, VDB_TEAM=4)
    print(f"   one query: single {t1*1e6:.0f} us, team 4 {t4*1e6:.0f} us")
    slow = int(np.argmax(rst["rows_scored"])); print(f"   slowest query {slow}: rows {rst['rows_scored'][slow]} exp {rst['expanded'][slow]}; median rows {np.median(rst['rows_scored'])}")
    for tw in (0, 2, 4):
        (_, _, s1), t = run(h, q[slow:slow+1], 10, ef, VDB_TEAM=tw)
        (_, _, s2), t2 = run(h, q[5:6], 10, ef, VDB_TEAM=tw)
        print(f"      team {tw}: slowest alone {t*1e6:.0f} us rows {s1['rows_scored'][0]}, query 5 alone {t2*1e6:.0f} us rows {s2['rows_scored'][0]}")
