"""GPU parity at sizes beyond the small fixtures: the graph the library builds for a 200k x 128 slice of the benchmark
workload against the CPU oracle (search parity on the same graph, builder quality against the oracle's point-by-point
build of hnsw.rs:119-208), and the sharded exchange (`ShardedIndex.merge_dev`) on the HIP backend."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, D, NQ, K = 200_000, 128, 2000, 10


def clustered(n, d, seed, centres, sigma=12.0):
    rng = np.random.default_rng(seed)
    c = centres[rng.integers(0, len(centres), size=n)]
    return np.clip(c + rng.standard_normal((n, d), dtype=np.float32) * sigma, 0, None).astype(np.float32)


def recall(ann, ann_n, exact, k):
    hits = 0
    for a, n, e in zip(ann, ann_n, exact):
        hits += len(set(a["id"][: min(n, k)].tolist()) & set(e["id"][:k].tolist()))
    return hits / (len(ann) * k)


@pytest.fixture(scope="module")
def workload(vdb, ora):
    """200k rows of the benchmark's generator (the same 4096-centre density: 820 centres), graph built by the library's
    batched inserter and by the oracle point by point."""
    centres = np.random.default_rng(0xC2).uniform(0, 128, size=(N * 4096 // 1_000_000, D)).astype(np.float32)
    base = clustered(N, D, 1, centres); queries = clustered(NQ, D, 2, centres)
    ids = np.arange(N, dtype=np.uint64) * 2 + 3
    M = vdb.DistanceMetric.Euclidean
    t0 = time.time()
    hb = vdb.HNSWIndex(M, 16, 64, 16, D, seed=7)
    hb.insert_batch(ids, base)
    tb = time.time() - t0
    ora.set_threads(min(os.cpu_count() or 1, 16))
    t0 = time.time()
    ho = ora.HNSWIndex(M, 16, 64, 16, D, seed=7)
    for lo in range(0, N, 25_000):                      # in pieces, with a progress line: a long silent run looks hung
        ho.insert_batch(ids[lo:lo + 25_000], base[lo:lo + 25_000])
        print(f"  oracle inserted {min(N, lo + 25_000)} ({time.time() - t0:.0f}s)", flush=True)
    print(f"batched GPU build {tb:.1f}s, sequential oracle build {time.time() - t0:.1f}s", flush=True)
    exact, _ = hb.exact_knn_batch(queries, K)
    return dict(base=base, queries=queries, ids=ids, hb=hb, ho=ho, exact=exact)


def test_batched_builder_recall_within_one_point_of_the_oracle_built_graph(vdb, workload):
    """Quality gate of the batched builder (rounds search a frozen graph, then link by hnsw.rs:175-205): recall@10 at
    ef 16 / 64 / 256 within 1 pt of a graph the oracle built with the reference's sequential insert, both searched by
    the library on the same queries."""
    w = workload
    hs = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=7)
    hs.import_graph(w["ho"].export_graph())
    for ef in (16, 64, 256):
        ob, cb = w["hb"].search_batch(w["queries"], K, ef_search=ef)
        os_, cs = hs.search_batch(w["queries"], K, ef_search=ef)
        rb, rs = recall(ob, cb, w["exact"], K), recall(os_, cs, w["exact"], K)
        print(f"  ef={ef}: batched {rb:.4f} sequential {rs:.4f}", flush=True)
        assert rb >= rs - 0.01, (ef, rb, rs)


def test_search_parity_on_the_library_built_graph(vdb, ora, workload):
    """The library's graph handed to the oracle; ef = 144 (the benchmark's).  STRICT: every id and key bit-identical,
    expansion counts equal (a key tie at the list's edge may change one walk).  FAST, one node per step: ids equal
    outside near-ties, keys within 1e-5 relative.  FAST default: recall within 0.5 pt of the oracle's."""
    w = workload
    q = w["queries"]; ef = 144
    g = w["hb"].export_graph()
    ho = ora.HNSWIndex(ora.DistanceMetric.Euclidean, 16, 64, 16, D); ho.import_graph(g)
    ref, refn, rst = ho.search_batch(q, K, ef_search=ef, with_stats=True)
    ps = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, arith=vdb.Arith.Strict); ps.import_graph(g)
    got, gotn, st = ps.search_batch(q, K, ef_search=ef, with_stats=True)
    assert np.array_equal(refn, gotn) and np.array_equal(ref["id"], got["id"])
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
    assert np.mean(rst["expanded"] == st["expanded"]) >= 0.999
    p1 = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, beam_width=1); p1.import_graph(g)
    got, gotn = p1.search_batch(q, K, ef_search=ef)
    assert np.array_equal(refn, gotn)
    rel = np.abs(ref["sort_key"].astype(np.float64) - got["sort_key"]) / np.maximum(ref["sort_key"], 1e-6)
    same = ref["id"] == got["id"]
    assert rel[same].max() <= 1e-5, rel[same].max()
    assert same.mean() >= 0.999, same.mean()
    got, gotn = w["hb"].search_batch(q, K, ef_search=ef)
    assert recall(got, gotn, w["exact"], K) >= recall(ref, refn, w["exact"], K) - 0.005


def test_exact_knn_at_this_size_matches_the_oracle_on_a_sample(ora, workload):
    """Ground truth of the recall figures: the MFMA + re-rank path against the oracle's brute force, 64 queries."""
    w = workload
    ho = w["ho"]
    ref, refn = ho.exact_knn_batch(w["queries"][:64], K)
    assert np.array_equal(ref["id"], w["exact"]["id"][:64])
    assert np.array_equal(ref["sort_key"].view(np.uint32), w["exact"]["sort_key"][:64].view(np.uint32))


def test_sharded_exchange_on_the_gpu_matches_one_index(vdb, workload):
    """Two shards (contiguous halves, each with its own index) on this GPU, the RCCL exchange step driven through
    `ShardedIndex.merge_dev(always_exchange=True)` in a process group of one rank, and the two-shard merge through the
    `gathered` path: the merged exact k-NN equals the exact k-NN of one index over the whole set, ids and keys bit for
    bit; a world of one returns its own lists unchanged."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from vectordb_amd.sharded import ShardedIndex, shard_range
    w = workload
    n = 60_000
    base, ids, q = w["base"][:n], w["ids"][:n], w["queries"][:500]
    M = vdb.DistanceMetric.Euclidean
    dev = torch.device("cuda", 0)
    shards = []
    for r in range(2):
        lo, hi = shard_range(n, 2, r)
        h = vdb.HNSWIndex(M, 16, 64, 16, D, seed=11 + r); h.insert_batch(ids[lo:hi], base[lo:hi]); shards.append(h)
    whole = vdb.HNSWIndex(M, 16, 64, 16, D, seed=11); whole.insert_batch(ids, base)
    want, want_n = whole.exact_knn_batch(q, K)

    def dev_lists(h, fn):
        out = torch.zeros(len(q) * K * 16, dtype=torch.uint8, device=dev); cnt = torch.zeros(len(q), dtype=torch.int32, device=dev)
        dq = torch.from_numpy(q).to(dev)
        torch.cuda.synchronize()
        vdb.check(fn(h._h, dq.data_ptr(), len(q), K, out.data_ptr(), cnt.data_ptr(), None))
        torch.cuda.synchronize()
        return out, cnt

    lists = [dev_lists(h, vdb.fn.index_exact_knn_batch_dev) for h in shards]
    store = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vdb_pg_{os.getpid()}")
    if os.path.exists(store): os.remove(store)
    dist.init_process_group("nccl", init_method=f"file://{store}", rank=0, world_size=1, device_id=dev)
    try:
        sh = ShardedIndex(vdb, shards[0], device=dev, always_exchange=True)
        assert sh.world == 1
        m_out, m_cnt = sh.merge_dev(lists[0][0], lists[0][1], len(q), K)     # all-gather over RCCL + merge kernel
        torch.cuda.synchronize()
        assert m_out.data_ptr() != lists[0][0].data_ptr(), "the exchange step must have run"
        assert torch.equal(m_out, lists[0][0]) and torch.equal(m_cnt, lists[0][1])
        g_out = torch.cat([lists[0][0], lists[1][0]]); g_cnt = torch.cat([lists[0][1], lists[1][1]])
        sh.world = 2                                                           # two gathered shards, merged on the GPU
        m_out, m_cnt = sh.merge_dev(g_out, g_cnt, len(q), K, gathered=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
        if os.path.exists(store): os.remove(store)
    got = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(len(q), K); got_n = m_cnt.cpu().numpy().view(np.uint32)
    assert np.array_equal(want_n, got_n) and np.array_equal(want["id"], got["id"])
    assert np.array_equal(want["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
    # the graph search of the two shards, merged, is the (sort_key, id) merge of their own lists
    a, an = shards[0].search_batch(q, K, ef_search=64); b, bn = shards[1].search_batch(q, K, ef_search=64)
    la = torch.from_numpy(np.concatenate([a.reshape(-1), b.reshape(-1)]).view(np.uint8)).to(dev)
    lc = torch.from_numpy(np.concatenate([an, bn]).view(np.int32)).to(dev)
    sh2 = ShardedIndex(vdb, shards[0], device=dev); sh2.world = 2
    m_out, m_cnt = sh2.merge_dev(la, lc, len(q), K, gathered=True)
    torch.cuda.synchronize()
    got = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(len(q), K)
    for i in range(len(q)):
        both = np.concatenate([a[i][: an[i]], b[i][: bn[i]]])
        o = np.lexsort((both["id"], both["sort_key"]))[:K]
        assert np.array_equal(both[o]["id"], got[i]["id"])


def test_shard_group_entry_points_in_a_group_of_one(vdb, workload):
    """The C-ABI shard group (one RCCL communicator created by the library, no torch.distributed): in a group of one
    rank the sharded index search, exact k-NN and filtered segment search — local search, all-gather, merge kernel on one
    stream — return exactly what the unsharded entry points return, and the communicator reports its own size."""
    torch = pytest.importorskip("torch")
    from vectordb_amd.sharded import ShardedIndex
    w = workload
    n, nq = 30_000, 300
    base, ids, q = w["base"][:n], w["ids"][:n], w["queries"][:nq]
    dev = torch.device("cuda", 0)
    h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=21); h.insert_batch(ids, base)
    sh = ShardedIndex(vdb, h, device=dev, always_exchange=True)
    assert sh.vgroup and sh.rccl_ranks() == 1 and vdb.fn.shard_group_rank(sh.vgroup) == 0
    dq = torch.from_numpy(q).to(dev)
    torch.cuda.synchronize()
    view = lambda o, c: (o.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, K), c.cpu().numpy().view(np.uint32))
    want, want_n = h.search_batch(q, K, ef_search=96)
    got, got_n = view(*sh.search_batch_dev(dq, nq, K, ef_search=96)); torch.cuda.synchronize()
    got, got_n = view(*sh.search_batch_dev(dq, nq, K, ef_search=96))
    assert np.array_equal(want_n, got_n) and np.array_equal(want, got)
    want, want_n = h.exact_knn_batch(q, K)
    got, got_n = view(*sh.exact_knn_batch_dev(dq, nq, K))
    assert np.array_equal(want_n, got_n) and np.array_equal(want, got)
    seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=22, filter_mode=vdb.FilterMode.Masked))
    sid = seg.insert_batch_columns(base[:8000], {"cat": (np.arange(8000) % 20).astype(np.int64)})
    for i in range(0, 8000, 37): seg.delete(int(sid[i]))
    for f in (None, vdb.Filter.Match("cat", vdb.PayloadValue.Int(3)), vdb.Filter.Compare("cat", vdb.ScalarComparisonOp.Lt, vdb.PayloadValue.Int(10))):
        want, want_n = seg.search_batch(q, K, ef_search=96, filter=f)
        got, got_n = view(*sh.segment_search_batch_dev(seg, dq, nq, K, ef_search=96, filter=f))
        assert np.array_equal(want_n, got_n)
        for i in range(nq): assert np.array_equal(want[i][: want_n[i]], got[i][: got_n[i]])
    # rows computed elsewhere go through the exchange alone
    lo = torch.from_numpy(want.view(np.uint8).reshape(-1).copy()).to(dev); lc = torch.from_numpy(want_n.view(np.int32).copy()).to(dev)
    m_out, m_cnt = sh.merge_dev(lo, lc, nq, K)
    torch.cuda.synchronize()
    assert m_out.data_ptr() != lo.data_ptr() and torch.equal(m_cnt, lc)
    mo = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, K)
    for i in range(nq): assert np.array_equal(mo[i][: want_n[i]], want[i][: want_n[i]])
