"""Size-independent properties at the benchmark's full size (1M x 128 fp32 L2, 10k queries, the generator of bench.py).

The oracle needs minutes for a graph of this size, so nothing here compares with it; the checks are the ones that hold
at any size: determinism (tests/hnsw.rs:376-390 of the reference: two identical searches are equal), ascending
sort keys and distinct ids (hnsw.rs:505-506, 550-551), every returned distance equal to a float64 recomputation of
metric.rs:30-35 within 1e-5 relative, exact k-NN results equal to a float64 brute force over a sample of the queries
(ids, outside 1e-6-relative ties), a stored row finds itself first at distance 0 in the exact search, recall@10 of the graph search against
the exact lists at the benchmark's ef, tombstoned rows never returned, and two half-size shards merged on the GPU equal
to the exact k-NN of the whole set bit for bit.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, D, Q, K, EF = 1_000_000, 128, 10_000, 10, 144


@pytest.fixture(scope="module")
def world(vdb):
    import bench
    centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, D)).astype(np.float32)
    base = bench.clustered(N, D, 0xC2 + 1, centres)
    queries = bench.clustered(Q, D, 0xC2 + 1000, centres)
    queries[:64] = base[7::15_000][:64]                      # stored rows as queries
    ids = np.arange(N, dtype=np.uint64) * 3 + 11             # ids are not slots
    h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=0xC2, arith=vdb.Arith.Fast)
    for s in range(0, N, 250_000):
        h.insert_batch(ids[s:s + 250_000], base[s:s + 250_000])
    return dict(base=base, queries=queries, ids=ids, h=h)


def _f64_dist(q, rows):
    d = rows.astype(np.float64) - q.astype(np.float64)[None, :]
    return np.sqrt((d * d).sum(axis=1))


def test_graph_search_properties_at_full_size(vdb, world):
    w = world
    h, base, queries, ids = w["h"], w["base"], w["queries"], w["ids"]
    out, cnt = h.search_batch(queries, K, ef_search=EF)
    out2, cnt2 = h.search_batch(queries, K, ef_search=EF)
    assert out.tobytes() == out2.tobytes() and cnt.tobytes() == cnt2.tobytes()          # determinism
    assert (cnt == K).all()
    assert (np.diff(out["sort_key"], axis=1) >= 0).all()                                 # ascending
    assert (out["raw_score"] == out["sort_key"]).all()                                   # L2: key = raw (hnsw.rs:103-108)
    srt = np.sort(out["id"], axis=1)
    assert (np.diff(srt.astype(np.int64), axis=1) != 0).all()                            # distinct ids per query
    assert ((out["id"] - 11) % 3 == 0).all() and (out["id"] < 3 * N + 11).all()
    # returned distances against float64, on a sample of the queries
    for qi in range(0, Q, 97):
        slots = ((out["id"][qi] - 11) // 3).astype(np.int64)
        ref = _f64_dist(queries[qi], base[slots])
        assert np.allclose(out["sort_key"][qi], ref, rtol=1e-5, atol=1e-30), qi
    # stored rows as queries: whenever the walk reaches the row it comes first at distance exactly 0 (the walk is
    # approximate: on this graph it reaches about as many of them as the recall says)
    found = 0
    for qi in range(64):
        if out["id"][qi][0] == ids[7 + 15_000 * qi]:
            assert out["sort_key"][qi][0] == 0.0
            found += 1
        else:
            assert ids[7 + 15_000 * qi] not in out["id"][qi]
    assert found >= 48
    # recall@10 against the exact lists at the benchmark's ef
    ex, exn = h.exact_knn_batch(queries, K)
    hits = 0
    for qi in range(Q):
        kth = ex["sort_key"][qi][K - 1]
        ok = np.isin(out["id"][qi], ex["id"][qi]) | (np.abs(out["sort_key"][qi] - kth) <= 1e-6 * max(abs(kth), 1e-30))
        hits += int(ok.sum())
    assert hits / (Q * K) >= 0.95
    w["exact"] = (ex, exn)


def test_exact_knn_equals_float64_brute_force_on_a_sample(vdb, world):
    w = world
    h, base, queries, ids = w["h"], w["base"], w["queries"], w["ids"]
    ex, exn = w.get("exact") or h.exact_knn_batch(queries, K)
    assert (exn == K).all() and (np.diff(ex["sort_key"], axis=1) >= 0).all()
    for qi in range(64):                                     # a stored row finds itself first, at distance exactly 0
        assert ex["sort_key"][qi][0] == 0.0 and ex["id"][qi][0] == ids[7 + 15_000 * qi]
    for qi in list(range(0, 64, 9)) + list(range(100, Q, 1111)):
        d = _f64_dist(queries[qi], base)
        order = np.lexsort((ids, d))[: K + 8]
        kth = d[order[K - 1]]
        for pos in range(K):
            got_slot = int((ex["id"][qi][pos] - 11) // 3)
            assert abs(float(ex["sort_key"][qi][pos]) - d[got_slot]) <= 1e-5 * max(d[got_slot], 1e-30)
            # the id is the float64 order's, or lies within a 1e-6-relative tie of it
            want = order[pos]
            assert got_slot == want or abs(d[got_slot] - d[want]) <= 1e-6 * max(d[want], 1e-30), (qi, pos)
        assert d[(ex["id"][qi] - 11) // 3].max() <= kth * (1 + 1e-6)


def test_tombstones_and_two_shards_at_full_size(vdb, world):
    import torch
    from vectordb_amd.sharded import ShardedIndex
    w = world
    h, base, queries, ids = w["h"], w["base"], w["queries"], w["ids"]
    qs = queries[:2000]
    ex, exn = h.exact_knn_batch(qs, K)
    # two shards of 500k rows each, merged by the GPU kernel: the exact k-NN of the whole set, bit for bit
    dev = torch.device("cuda", 0)
    parts = []
    for lo, hi in ((0, N // 2), (N // 2, N)):
        s = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=5, arith=vdb.Arith.Fast)
        s.insert_batch(ids[lo:hi], base[lo:hi])
        parts.append(s.exact_knn_batch(qs, K))
        del s
    g_out = torch.from_numpy(np.concatenate([p[0].view(np.uint8).reshape(-1) for p in parts])).to(dev)
    g_cnt = torch.from_numpy(np.concatenate([p[1].view(np.int32) for p in parts])).to(dev)
    sh = ShardedIndex(vdb, h, device=dev)
    sh.world = 2                                             # two gathered shards in one process
    m_out, m_cnt = sh.merge_dev(g_out, g_cnt, len(qs), K, gathered=True)
    torch.cuda.synchronize()
    mo = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(len(qs), K)
    assert mo.tobytes() == ex.tobytes()
    # tombstones: delete every query's best row, it never comes back and the runner-up moves up
    best = np.unique(ex["id"][:, 0])
    for i in best:
        h.mark_deleted(int(i))
    out, cnt = h.search_batch(qs, K, ef_search=EF)
    assert not np.isin(out["id"], best).any()
    ex2, _ = h.exact_knn_batch(qs, K)
    assert not np.isin(ex2["id"], best).any()
    keep = ~np.isin(ex["id"][:, 1], best)
    assert (ex2["id"][keep, 0] == ex["id"][keep, 1]).all()
