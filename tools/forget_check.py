"""Diagnostic: the register beam must return the same lists whatever the size of its visited table."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered
import vectordb_amd
vdb = vectordb_amd.api()
n, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128, 10000, 10
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 144
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, d)).astype(np.float32)
base = clustered(n, d, 0xC3, centres); q = clustered(nq, d, 0xC2 + 1000, centres)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d, seed=0xC2)
h.insert_batch(np.arange(n, dtype=np.uint64), base)
res = {}
for t in (15, 13, 12, 11, 9, 7):
    os.environ["VDB_TAG_LOG2"] = str(t)
    out, cnt, st = h.search_batch(q, k, ef_search=ef, with_stats=True)
    res[t] = (out.copy(), cnt.copy(), st.copy())
    dups = sum(len(set(out["id"][i][:cnt[i]].tolist())) != cnt[i] for i in range(nq))
    print(f"tags {t}: rows_scored {st['rows_scored'].mean():.1f}, queries with a repeated id {dups}")
ref = res[15]
for t in (13, 12, 11, 9, 7):
    o, c, s = res[t]
    bad = np.nonzero((o["id"] != ref[0]["id"]).any(axis=1))[0]
    print(f"tags {t} vs 15: {len(bad)} queries differ; expansions equal: {np.array_equal(s['expanded'], ref[2]['expanded'])}")
    for i in bad[:3]:
        print("  q", i, "ref", ref[0]["id"][i].tolist(), ref[0]["sort_key"][i].tolist()); print("       got", o["id"][i].tolist(), o["sort_key"][i].tolist())
