"""Device-side filter compiler: the pass / err bitmaps the GPU computes from the payload columns and the inverted index
against evaluate_filter (filters.rs:23-75, payload.rs:36-186) applied to every payload by the CPU oracle, for every
value kind, every comparison, type mismatches, missing fields, list fields, Err propagation through And / Or / Not,
after further inserts, deletes and a purge; and the device-buffer filtered search against the host-buffer one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def payload_spec(i):
    """Field values of point i (None: the point has no payload at all)."""
    if i % 11 == 10:
        return None
    spec = {"cat": ("Int", i % 7), "tier": ("Int", (i * 2654435761) % 5)}
    if i % 3 != 0: spec["score"] = ("Float", [float(i % 13) - 6.0, float("nan"), -0.0, 0.0, 2.5][i % 5])
    if i % 4 != 1: spec["name"] = ("Str", ["alpha", "beta", "gamma", "", "zeta"][i % 5])
    if i % 5 != 2: spec["flag"] = ("Bool", i % 2 == 0)
    if i % 6 == 0: spec["tags"] = ("ListStr", [["red", "blue"], ["green"], []][(i // 6) % 3])
    if i % 6 == 3: spec["tags"] = ("Str", "red")                      # the same key holding another kind
    if i % 9 == 4: spec["nums"] = ("ListInt", [i % 3, 7])
    if i % 10 == 7: spec["cat"] = ("Float", 3.0)                       # Int column with a few Float values
    return spec


def make_payload(api, spec):
    p = api.Payload()
    for k, (kind, v) in spec.items():
        p.set(k, getattr(api.PayloadValue, kind)(v))
    return p


def filters(api):
    V, F, Op = api.PayloadValue, api.Filter, api.ScalarComparisonOp
    ops = [Op.Eq, Op.Neq, Op.Lt, Op.Lte, Op.Gt, Op.Gte]
    out = []
    out += [("match cat=3", F.Match("cat", V.Int(3))), ("match cat=3.0", F.Match("cat", V.Float(3.0))), ("match tier=9", F.Match("tier", V.Int(9))),
            ("match score=nan", F.Match("score", V.Float(float("nan")))), ("match score=0", F.Match("score", V.Float(0.0))),
            ("match name=beta", F.Match("name", V.Str("beta"))), ("match name=''", F.Match("name", V.Str(""))), ("match name=nope", F.Match("name", V.Str("nope"))),
            ("match flag", F.Match("flag", V.Bool(True))), ("match missing key", F.Match("ghost", V.Int(1))),
            ("match tags=[green]", F.Match("tags", V.ListStr(["green"]))), ("match tags=red", F.Match("tags", V.Str("red"))),
            ("match nums", F.Match("nums", V.ListInt([1, 7])))]
    for op in ops:
        out += [(f"cmp cat {op.name} 3", F.Compare("cat", op, V.Int(3))), (f"cmp score {op.name} 0", F.Compare("score", op, V.Float(0.0))),
                (f"cmp score {op.name} nan", F.Compare("score", op, V.Float(float("nan")))), (f"cmp name {op.name} beta", F.Compare("name", op, V.Str("beta"))),
                (f"cmp flag {op.name}", F.Compare("flag", op, V.Bool(False))), (f"cmp tags {op.name} red", F.Compare("tags", op, V.Str("red"))),
                (f"cmp tags {op.name} list", F.Compare("tags", op, V.ListStr(["green"]))), (f"cmp nums {op.name}", F.Compare("nums", op, V.ListInt([1, 7])))]
    out += [("cmp type mismatch", F.Compare("cat", Op.Eq, V.Str("3"))), ("cmp ghost", F.Compare("ghost", Op.Lt, V.Int(5)))]
    a, b, c = (lambda: F.Match("cat", V.Int(2))), (lambda: F.Compare("score", Op.Gte, V.Float(0.0))), (lambda: F.Compare("name", Op.Lt, V.Str("c")))
    out += [("and", F.And([a(), b()])), ("and err first", F.And([b(), a()])), ("or", F.Or([a(), b()])), ("or err first", F.Or([b(), a()])),
            ("not", F.Not(a())), ("not err", F.Not(b())), ("and of three", F.And([F.Match("flag", V.Bool(True)), c(), F.Compare("tier", Op.Neq, V.Int(0))])),
            ("nested", F.Or([F.And([a(), F.Not(c())]), F.Not(F.Or([F.Match("tier", V.Int(1)), b()]))])),
            ("empty and", F.And([])), ("empty or", F.Or([])), ("not empty or", F.Not(F.Or([])))]
    return out


def expected(ora, specs, deleted):
    """Per slot: (pass, err) from the oracle's evaluate_filter on that point's payload."""
    res = {}
    pay = [None if s is None else make_payload(ora, s) for s in specs]
    for name, f in filters(ora):
        p = np.zeros(len(specs), bool); e = np.zeros(len(specs), bool)
        for i, pl in enumerate(pay):
            if pl is None: continue
            try: p[i] = ora.evaluate_filter(f, pl)
            except ora.InvalidPayload: e[i] = True
        res[name] = (p, e)
    return res


def check(seg, vdb, want, live):
    for name, f in filters(vdb):
        p, e = seg.filter_bitmaps(f)
        wp, we = want[name]
        assert np.array_equal(p[live], wp[live]), (name, "pass", np.nonzero(p[live] != wp[live])[0][:5])
        assert np.array_equal(e[live], we[live]), (name, "err", np.nonzero(e[live] != we[live])[0][:5])
        n_pass, n_err = seg.compile_filter(f)
        assert (n_pass, n_err) == (int(wp[live].sum()), int(we[live].sum())), name


@pytest.mark.parametrize("path", ["auto", "postings", "column"])
def test_device_bitmaps_equal_evaluate_filter_on_every_payload(ora, vdb, monkeypatch, path):
    if path != "auto": monkeypatch.setenv("VDB_FILTER_MATCH", path)
    n, d = 1500, 8
    rng = np.random.default_rng(17)
    x = rng.normal(size=(n + 400, d)).astype(np.float32)
    specs = [payload_spec(i) for i in range(n + 400)]
    seg = vdb.Segment(vdb.HNSWIndex(2, 8, 32, 16, d, seed=1, filter_mode=vdb.FilterMode.Masked))
    ids = seg.insert_batch(x[:n], [None if s is None else make_payload(vdb, s) for s in specs[:n]])
    live = np.ones(n, bool)
    check(seg, vdb, expected(ora, specs[:n], None), live)
    # deletes (tombstones leave the counts), then more points one by one and in a batch: the columns grow on the device
    for i in range(0, n, 17):
        seg.delete(int(ids[i])); live[i] = False
    for i in range(n, n + 100):
        seg.insert(x[i], None if specs[i] is None else make_payload(vdb, specs[i]))
    seg.insert_batch(x[n + 100:], [None if s is None else make_payload(vdb, s) for s in specs[n + 100:]])
    live = np.concatenate([live, np.ones(400, bool)])
    check(seg, vdb, expected(ora, specs, None), live)


def test_bitmaps_after_a_purge_follow_the_new_slots(ora, vdb):
    n, d = 600, 6
    rng = np.random.default_rng(3)
    x = rng.normal(size=(n, d)).astype(np.float32)
    specs = [payload_spec(i) for i in range(n)]
    seg = vdb.Segment(vdb.HNSWIndex(2, 8, 32, 16, d, seed=1, filter_mode=vdb.FilterMode.Masked))
    ids = seg.insert_batch(x, [None if s is None else make_payload(vdb, s) for s in specs])
    gone = list(range(0, n, 3))                               # a third of the points: the 100th delete purges (segment.rs:88-95)
    for i in gone[:150]:
        seg.delete(int(ids[i]))
    assert seg.hnsw().len() < n, "the deletes are meant to trigger a purge"
    left = [i for i in range(n) if not seg.is_deleted(int(ids[i])) and seg.hnsw().contains(int(ids[i]))]
    order = sorted(left, key=lambda i: int(ids[i]))           # after a purge slots follow ascending id
    kept = [specs[i] for i in order]
    assert seg.hnsw().len() == len(kept)
    check(seg, vdb, expected(ora, kept, None), np.ones(len(kept), bool))


@pytest.mark.parametrize("metric", [1, 2])
def test_filtered_search_on_device_buffers_equals_the_host_buffer_call(vdb, metric):
    torch = pytest.importorskip("torch")
    n, d, nq, k = 6000, 32, 200, 10
    rng = np.random.default_rng(23)
    x = rng.normal(size=(n, d)).astype(np.float32); q = rng.normal(size=(nq, d)).astype(np.float32)
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 50)); p.set("half", vdb.PayloadValue.Int(i % 2)); pay.append(p)
    seg = vdb.Segment(vdb.HNSWIndex(metric, 16, 64, 16, d, seed=2, filter_mode=vdb.FilterMode.Masked))
    ids = seg.insert_batch(x, pay)
    for i in range(0, n, 29): seg.delete(int(ids[i]))
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    V, F, Op = vdb.PayloadValue, vdb.Filter, vdb.ScalarComparisonOp
    cases = [None, F.Match("cat", V.Int(7)),                                              # 2 %: scan of the passing rows, read in place
             F.Match("half", V.Int(1)),                                                     # 50 %: masked walk
             F.And([F.Match("half", V.Int(0)), F.Compare("cat", Op.Lt, V.Int(10))]),         # 10 %
             F.Match("cat", V.Int(777))]                                                     # nothing passes
    for f in cases:
        want, want_n = seg.search_batch(q, k, ef_search=96, filter=f)
        out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); cnt = torch.full((nq,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        vdb.check(vdb.fn.segment_search_batch_dev(seg._h, dq.data_ptr(), nq, k, 96, f._h if f else None, out.data_ptr(), cnt.data_ptr(), None, None))
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, k); got_n = cnt.cpu().numpy().view(np.uint32)
        assert np.array_equal(want_n, got_n)
        for i in range(nq):
            assert np.array_equal(want["id"][i][: want_n[i]], got["id"][i][: got_n[i]])
            assert np.array_equal(want["sort_key"][i][: want_n[i]].view(np.uint32), got["sort_key"][i][: got_n[i]].view(np.uint32))
    lit = vdb.Segment(vdb.HNSWIndex(metric, 16, 64, 16, d, seed=2))          # literal mode: refused with a filter, fine without
    lit.insert_batch(x[:500], pay[:500])
    out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    with pytest.raises(vdb.ArgumentError, match="MASKED"):
        vdb.check(vdb.fn.segment_search_batch_dev(lit._h, dq.data_ptr(), nq, k, 0, cases[1]._h, out.data_ptr(), cnt.data_ptr(), None, None))
    vdb.check(vdb.fn.segment_search_batch_dev(lit._h, dq.data_ptr(), nq, k, 0, None, out.data_ptr(), cnt.data_ptr(), None, None))
    torch.cuda.synchronize()


def test_more_streams_and_filters_than_the_library_remembers(vdb):
    """24 filters on 24 streams queued back to back without a host synchronisation: the segment's cache keeps 16 compiled
    filters and an index 16 streams' scratch, so entries are dropped while launches that read them are still queued."""
    torch = pytest.importorskip("torch")
    n, d, nq, k = 5000, 32, 300, 10
    rng = np.random.default_rng(41)
    x = rng.normal(size=(n, d)).astype(np.float32); q = rng.normal(size=(nq, d)).astype(np.float32)
    seg = vdb.Segment(vdb.HNSWIndex(2, 16, 64, 16, d, seed=2, filter_mode=vdb.FilterMode.Masked))
    seg.insert_batch_columns(x, {"cat": (np.arange(n) % 48).astype(np.int64), "half": (np.arange(n) % 2).astype(np.int64)})
    V, F, Op = vdb.PayloadValue, vdb.Filter, vdb.ScalarComparisonOp
    filters = [F.Match("cat", V.Int(c)) if c % 2 else F.And([F.Match("half", V.Int(0)), F.Compare("cat", Op.Lt, V.Int(c + 2))]) for c in range(24)]
    want = [seg.search_batch(q, k, ef_search=64, filter=f) for f in filters]
    want_knn = seg.hnsw().exact_knn_batch(q, k)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in filters]
    outs = [torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev) for _ in filters]; cnts = [torch.full((nq,), -1, dtype=torch.int32, device=dev) for _ in filters]
    kouts = [torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev) for _ in filters]; kcnts = [torch.full((nq,), -1, dtype=torch.int32, device=dev) for _ in filters]
    torch.cuda.synchronize()
    for rnd in range(2):
        for f, st, o, c, ko, kc in zip(filters, streams, outs, cnts, kouts, kcnts):
            vdb.check(vdb.fn.segment_search_batch_dev(seg._h, dq.data_ptr(), nq, k, 64, f._h, o.data_ptr(), c.data_ptr(), None, st.cuda_stream))
            vdb.check(vdb.fn.index_exact_knn_batch_dev(seg.hnsw()._h, dq.data_ptr(), nq, k, ko.data_ptr(), kc.data_ptr(), st.cuda_stream))
    torch.cuda.synchronize()
    for (w, wn), o, c, ko, kc in zip(want, outs, cnts, kouts, kcnts):
        got = o.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, k); got_n = c.cpu().numpy().view(np.uint32)
        assert np.array_equal(wn, got_n)
        for i in range(nq):
            assert np.array_equal(w["id"][i][: wn[i]], got["id"][i][: wn[i]])
            assert np.array_equal(w["sort_key"][i][: wn[i]].view(np.uint32), got["sort_key"][i][: wn[i]].view(np.uint32))
        gk = ko.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, k)
        assert np.array_equal(kc.cpu().numpy().view(np.uint32), want_knn[1])
        assert np.array_equal(gk["id"], want_knn[0]["id"])


def test_payload_columns_build_the_same_segment_as_payload_objects(vdb):
    n, d = 2000, 16
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, d)).astype(np.float32)
    cat = (np.arange(n) % 13).astype(np.int64); w = (np.arange(n) % 4).astype(np.float64) * 0.5; odd = np.arange(n) % 2 == 1
    def seg(): return vdb.Segment(vdb.HNSWIndex(2, 8, 32, 16, d, seed=6, filter_mode=vdb.FilterMode.Masked))
    a, b = seg(), seg()
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(int(cat[i]))); p.set("w", vdb.PayloadValue.Float(float(w[i]))); p.set("odd", vdb.PayloadValue.Bool(bool(odd[i]))); pay.append(p)
    ia = a.insert_batch(x, pay); ib = b.insert_batch_columns(x, {"cat": cat, "w": w, "odd": odd})
    assert np.array_equal(ia, ib)
    ga, gb = a.hnsw().export_graph(), b.hnsw().export_graph()
    for (o1, f1, n1), (o2, f2, n2) in zip(ga["layers"], gb["layers"]):
        assert np.array_equal(o1, o2) and np.array_equal(f1, f2) and np.array_equal(n1, n2)
    f = vdb.Filter.And([vdb.Filter.Match("odd", vdb.PayloadValue.Bool(True)), vdb.Filter.Compare("w", vdb.ScalarComparisonOp.Gte, vdb.PayloadValue.Float(1.0))])
    assert a.compile_filter(f) == b.compile_filter(f) == (int(((np.arange(n) % 2 == 1) & (w >= 1.0)).sum()), 0)
    pa, ea = a.filter_bitmaps(f); pb, eb = b.filter_bitmaps(f)
    assert np.array_equal(pa, pb) and not ea.any() and not eb.any()
