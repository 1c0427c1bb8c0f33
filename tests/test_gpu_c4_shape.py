"""The embedding-shape configuration at reduced size: 768-d unit rows, dot product, payload {cat: i mod 100,
tier: mix(i) mod 10} ingested column-wise, filters at 1 % / 5 % / 10 % / 50 % selectivity in Masked mode
(hnsw.rs:585-689 with the filter as a candidate mask, stores.rs:55-61 for the Match leaves).

Narrow filters take the scan of the passing rows in place (long-row matrix-core candidate pass reading rows through the
row map, then the strict re-rank): checked bit for bit against the oracle's brute force over the live passing rows.
The 50 % filter walks the graph (64 lanes per 768-float row): STRICT against the oracle's masked walk on the same
segment, FAST by recall against the brute force over the passing rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, K = 768, 10
REL, TIE = 1e-5, 1e-6


def unit_rows(n, seed):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(0xC4).uniform(0, 128, size=(64, D)).astype(np.float32)
    x = np.clip(centres[rng.integers(0, len(centres), size=n)] + rng.standard_normal((n, D), dtype=np.float32) * 12.0, 0, None)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def columns(n):
    i = np.arange(n, dtype=np.uint64)
    cat = (i % np.uint64(100)).astype(np.int64)
    tier = (((i * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(10)).astype(np.int64)
    return cat, tier


def cases(api, cat, tier):
    PV, F, Op = api.PayloadValue, api.Filter, api.ScalarComparisonOp
    return [("cat == 7 (1 %)", F.Match("cat", PV.Int(7)), cat == 7),
            ("tier == 3 and cat < 50 (5 %)", F.And([F.Match("tier", PV.Int(3)), F.Compare("cat", Op.Lt, PV.Int(50))]), (tier == 3) & (cat < 50)),
            ("tier == 3 (10 %)", F.Match("tier", PV.Int(3)), tier == 3)]


def brute_force(ora, x, ids, rows, q, k):
    """The oracle's exact k-NN over an index that holds only `rows` (no graph: the brute force reads rows only)."""
    g = dict(ids=ids[rows], levels=np.zeros(len(rows), np.uint32), vectors=x[rows], deleted=np.zeros(len(rows), np.uint8),
             layers=[], entry=int(ids[rows[0]]), current_max_level=0)
    ho = ora.HNSWIndex(ora.DistanceMetric.Dot, 16, 64, 16, D); ho.import_graph(g)
    return ho.exact_knn_batch(q, k)


def on_device_buffers(vdb, seg, q, k, ef, f):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    nq = len(q)
    dq = torch.from_numpy(q).to(dev)
    out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); cnt = torch.full((nq,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    vdb.check(vdb.fn.segment_search_batch_dev(seg._h, dq.data_ptr(), nq, k, ef, f._h, out.data_ptr(), cnt.data_ptr(), None, None))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, k), cnt.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("arith,n,nq", [("fast", 60_000, 300), ("strict", 8_000, 40)])
def test_narrow_filters_scan_the_passing_768d_rows_like_the_brute_force(ora, vdb, arith, n, nq):
    ar = vdb.Arith.Fast if arith == "fast" else vdb.Arith.Strict
    x = unit_rows(n, 1); q = unit_rows(nq, 2)
    cat, tier = columns(n)
    seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Dot, 16, 64, 16, D, seed=0xC4, arith=ar, filter_mode=vdb.FilterMode.Masked))
    ids = np.concatenate([seg.insert_batch_columns(x[lo:lo + 25_000], {"cat": cat[lo:lo + 25_000], "tier": tier[lo:lo + 25_000]})
                          for lo in range(0, n, 25_000)])
    assert np.array_equal(ids, np.arange(1, n + 1, dtype=np.uint64))           # segment.rs:27,33
    live = np.ones(n, bool)
    for i in list(range(7, n, 1900)) + list(range(3, n, 2111)):                # tombstones among the passing rows and elsewhere
        seg.delete(int(ids[i])); live[i] = False
    for name, f, mask in cases(vdb, cat, tier):
        rows = np.nonzero(mask & live)[0]
        assert seg.compile_filter(f) == (len(rows), 0), name
        ref, refn = brute_force(ora, x, ids, rows, q, K)
        got, gotn = seg.search_batch(q, K, ef_search=128, filter=f)
        assert np.array_equal(refn, gotn), name
        assert np.array_equal(ref["id"], got["id"]), (name, np.argwhere(ref["id"] != got["id"])[:5])
        assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32)), name
        assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32)), name
        dev, devn = on_device_buffers(vdb, seg, q, K, 128, f)
        assert np.array_equal(gotn, devn) and np.array_equal(got, dev), name
        one = seg.search_with_filter(q[1], K, f)                               # segment.rs:121-147, one query per call
        assert [r.id for r in one] == got["id"][1][: gotn[1]].tolist()


def payload_segment(api, x, cat, tier, **kw):
    seg = api.Segment(api.HNSWIndex(api.DistanceMetric.Dot, 16, 64, 16, D, seed=0xC4, filter_mode=api.FilterMode.Masked, **kw))
    for i in range(len(x)):
        p = api.Payload(); p.set("cat", api.PayloadValue.Int(int(cat[i]))); p.set("tier", api.PayloadValue.Int(int(tier[i])))
        seg.insert(x[i], p)
    return seg


def recall(ann, ann_n, exact, exact_n, k):
    return sum(len(set(a["id"][:an].tolist()) & set(e["id"][:en].tolist())) for a, an, e, en in zip(ann, ann_n, exact, exact_n)) / (len(ann) * k)


def test_half_of_the_rows_pass_the_masked_walk_over_768d_rows(ora, vdb, monkeypatch, tmp_path):
    monkeypatch.setenv("VDB_MASKED_SCAN", "0")                                # the walk, whatever the share of passing rows
    n, nq, ef = 4000, 120, 128
    x = unit_rows(n, 11); q = unit_rows(nq, 12)
    cat, tier = columns(n)
    so = payload_segment(ora, x, cat, tier); so.set_filter_mode(ora.FilterMode.Masked)
    sp = payload_segment(vdb, x, cat, tier, arith=vdb.Arith.Strict)
    # hnsw.rs:119-325 on both sides builds the same lists except where two candidates of an insert score exactly the
    # same (dot products of unit rows near 1 share few fp32 values; the order of equal keys is BinaryHeap-internal in
    # the reference): the walk is compared on ONE graph, the library's, handed to the oracle's segment
    go, gp = so.hnsw().export_graph(), sp.hnsw().export_graph()
    assert np.array_equal(go["levels"], gp["levels"]) and np.array_equal(go["layers"][0][1], gp["layers"][0][1])   # same degrees
    assert np.mean(go["layers"][0][2] == gp["layers"][0][2]) > 0.999
    so.hnsw().import_graph(gp)
    for i in range(5, n, 173):
        so.delete(i + 1); sp.delete(i + 1)
    half = lambda a: a.Filter.Compare("cat", a.ScalarComparisonOp.Lt, a.PayloadValue.Int(50))
    ref, refn = so.search_batch(q, K, ef_search=ef, filter=half(ora))
    got, gotn = sp.search_batch(q, K, ef_search=ef, filter=half(vdb))
    assert np.array_equal(refn, gotn)
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
    assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32))
    for qi in np.nonzero((ref["id"] != got["id"]).any(axis=1))[0]:           # ids may only trade places inside a run of equal keys
        for j in np.nonzero(ref["id"][qi] != got["id"][qi])[0]:
            tied = ref["sort_key"][qi] == ref["sort_key"][qi][j]
            assert tied.sum() > 1 and got["id"][qi][j] in ref["id"][qi][tied], (qi, j)
    assert np.mean(ref["id"] == got["id"]) > 0.995
    dev, devn = on_device_buffers(vdb, sp, q, K, ef, half(vdb))
    assert np.array_equal(gotn, devn) and np.array_equal(got, dev)
    for qi in range(3):                                                        # one query per call, at the index's own ef
        a = so.in_place_filtered_search(q[qi], K, half(ora)); b = sp.in_place_filtered_search(q[qi], K, half(vdb))
        assert [np.float32(r.sort_key).view(np.uint32) for r in a] == [np.float32(r.sort_key).view(np.uint32) for r in b]
    # FAST arithmetic on the same segment (saved and loaded): 64 lanes per row on the register beam
    path = tmp_path / "c4.seg"; sp.save(path)
    live = np.array([not so.is_deleted(i + 1) for i in range(n)])
    rows = np.nonzero((cat < 50) & live)[0]
    ids = np.arange(1, n + 1, dtype=np.uint64)
    exact, exact_n = brute_force(ora, x, ids, rows, q, K)
    passing = set(int(ids[r]) for r in rows)
    for width in (1, 0):
        sf = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Dot, 16, 64, 16, D, seed=0xC4, filter_mode=vdb.FilterMode.Masked,
                                       arith=vdb.Arith.Fast, beam_width=width))
        sf.load(path)
        fast, fastn = sf.search_batch(q, K, ef_search=ef, filter=half(vdb))
        assert np.array_equal(refn, fastn)
        assert set(fast["id"].ravel().tolist()) <= passing
        assert recall(fast, fastn, exact, exact_n, K) >= recall(ref, refn, exact, exact_n, K) - 0.005
        devf, devfn = on_device_buffers(vdb, sf, q, K, ef, half(vdb))
        assert np.array_equal(fastn, devfn) and np.array_equal(fast, devf)
        if width == 1:                                                         # the reference's order of expansion: same ids outside near-ties
            bad = 0
            for qi in range(nq):
                r, g = ref[qi][: refn[qi]], fast[qi][: fastn[qi]]
                # dot products of unit rows: the 1e-5 bar is relative to |q||x| = 1
                assert np.all(np.abs(r["sort_key"].astype(np.float64) - g["sort_key"].astype(np.float64)) <= REL)
                for j in np.nonzero(r["id"] != g["id"])[0]:
                    near = np.abs(r["sort_key"].astype(np.float64) - float(r["sort_key"][j])) <= TIE
                    bad += g["id"][j] not in r["id"][near]
            assert bad <= nq * K // 200, bad
