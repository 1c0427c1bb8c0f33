"""Graph file (layout: include/vdbmi.h): the Python reader/writer against the oracle on CPU, and on the GPU box
the C entry points vdb_index_save / vdb_index_load against it in both directions."""
import numpy as np
import pytest

from vectordb_amd import graph_file


def _build(api, metric, n, d, seed, **kw):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    h = api.HNSWIndex(metric, 8, 32, 16, d, seed=seed, **kw)
    for i in range(n):
        h.insert(i * 7 + 3, x[i])
    h.mark_deleted(3 + 7 * 5)
    return h, x


def _same_graph(a, b):
    assert a["entry"] == b["entry"] and a["current_max_level"] == b["current_max_level"]
    for key in ("ids", "levels", "deleted"):
        assert np.array_equal(a[key], b[key])
    assert np.array_equal(a["vectors"].view(np.uint32), b["vectors"].view(np.uint32))
    assert len(a["layers"]) == len(b["layers"])
    for la, lb in zip(a["layers"], b["layers"]):
        for u, v in zip(la, lb):
            assert np.array_equal(u, v)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_oracle_graph_round_trips_through_the_file(ora, tmp_path, metric):
    h, x = _build(ora, metric, 300, 12, 5 + metric)
    path = tmp_path / "g.vdbgraf"
    h.save(path)
    cfg, g = graph_file.read_graph(path)
    assert (cfg["metric"], cfg["m"], cfg["ef"], cfg["max_level_cap"], cfg["dim"]) == (metric, 8, 32, 16, 12)
    _same_graph(h.export_graph(), g)
    h2 = ora.HNSWIndex(metric, 8, 32, 16, 12, seed=1)
    h2.load(path)
    q = x[:20] + 0.01
    a, an = h.search_batch(q, 5); b, bn = h2.search_batch(q, 5)
    assert np.array_equal(an, bn) and np.array_equal(a["id"], b["id"]) and np.array_equal(a["sort_key"].view(np.uint32), b["sort_key"].view(np.uint32))


def test_empty_index_and_bad_files(ora, tmp_path):
    h = ora.HNSWIndex(2, 8, 32, 16, 4)
    path = tmp_path / "empty.vdbgraf"
    h.save(path)
    cfg, g = graph_file.read_graph(path)
    assert len(g["ids"]) == 0 and g["entry"] is None and g["vectors"].shape == (0, 4)
    raw = path.read_bytes()
    (tmp_path / "short").write_bytes(raw[:50])
    with pytest.raises(ValueError):
        graph_file.read_graph(tmp_path / "short")
    (tmp_path / "magic").write_bytes(b"NOTAGRAF" + raw[8:])
    with pytest.raises(ValueError):
        graph_file.read_graph(tmp_path / "magic")
    other = ora.HNSWIndex(2, 8, 32, 16, 5)
    with pytest.raises(ora.DBError):
        other.load(path)                                   # dimension 4 file into a dimension 5 index


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_library_and_python_files_are_interchangeable(ora, vdb, tmp_path, metric):
    ho, x = _build(ora, metric, 400, 20, 15 + metric)
    q = x[:30] + 0.02
    ref, refn = ho.search_batch(q, 10)
    # oracle graph, written by the Python writer, loaded by the library
    p1 = tmp_path / "from_oracle.vdbgraf"; ho.save(p1)
    hp = vdb.HNSWIndex(metric, 8, 32, 16, 20, seed=2, arith=vdb.Arith.Strict)
    hp.load(p1)
    got, gotn = hp.search_batch(q, 10)
    assert np.array_equal(refn, gotn) and np.array_equal(ref["id"], got["id"]) and np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
    # written by the library: identical arrays, readable by the Python reader and by a second library index
    p2 = tmp_path / "from_library.vdbgraf"; hp.save(p2)
    cfg, g = graph_file.read_graph(p2)
    _same_graph(ho.export_graph(), g)
    assert cfg["seed"] == 0 and cfg["dim"] == 20               # load took over the seed of the file it read (the Python writer stores 0)
    hq = vdb.HNSWIndex(metric, 8, 32, 16, 20, seed=2, arith=vdb.Arith.Strict)
    hq.load(p2)
    again, againn = hq.search_batch(q, 10)
    assert np.array_equal(got["id"], again["id"]) and np.array_equal(gotn, againn)
    # a built (not imported) library index keeps its random-stream positions across save/load: the next insert draws the same level
    hb, xb = _build(vdb, metric, 200, 20, 33, arith=vdb.Arith.Strict)
    p3 = tmp_path / "built.vdbgraf"; hb.save(p3)
    hc = vdb.HNSWIndex(metric, 8, 32, 16, 20, seed=99, arith=vdb.Arith.Strict); hc.load(p3)     # the file's seed replaces 99
    extra = np.random.default_rng(1).uniform(-1, 1, size=(40, 20)).astype(np.float32)
    for i in range(40):
        hb.insert(10_000 + i, extra[i]); hc.insert(10_000 + i, extra[i])
    _same_graph(hb.export_graph(), hc.export_graph())
    with pytest.raises(vdb.DBError):
        vdb.HNSWIndex(metric, 8, 32, 16, 21).load(p3)


@pytest.mark.gpu
def test_bad_graphs_and_files_are_rejected_and_leave_the_index_untouched(ora, vdb, tmp_path):
    """vdb_index_import_graph / vdb_index_load validate before they touch the index: every malformed CSR is
    VDB_ERR_ARGUMENT, an unreadable file VDB_ERR_IO, and the index answers exactly as before afterwards."""
    ho, x = _build(ora, 2, 300, 12, 41)
    g = ho.export_graph()
    hp = vdb.HNSWIndex(2, 8, 32, 16, 12, arith=vdb.Arith.Strict); hp.import_graph(g)
    q = x[:25] + 0.01
    ref, refn = hp.search_batch(q, 5)

    def broken(edit):
        b = dict(g); b["layers"] = [tuple(a.copy() for a in l) for l in g["layers"]]
        b["ids"] = g["ids"].copy()
        edit(b)
        return b

    def swap_offsets(b): b["layers"][0][1][3], b["layers"][0][1][4] = b["layers"][0][1][4] + 5, b["layers"][0][1][3]
    def short_cover(b): b["layers"][0][1][-1] -= 1
    def first_offset(b): b["layers"][0][1][0] = 1
    def owner_range(b): b["layers"][0][0][2] = 300
    def owner_twice(b): b["layers"][0][0][2] = b["layers"][0][0][1]
    def neighbour_range(b): b["layers"][0][2][7] = 300
    def id_twice(b): b["ids"][9] = b["ids"][8]
    def entry_unknown(b): b["entry"] = 999_999
    for edit in (swap_offsets, short_cover, first_offset, owner_range, owner_twice, neighbour_range, id_twice, entry_unknown):
        with pytest.raises(vdb.ArgumentError):
            hp.import_graph(broken(edit))
        got, gotn = hp.search_batch(q, 5)
        assert np.array_equal(refn, gotn) and np.array_equal(ref["id"], got["id"]), edit.__name__
    # files: truncated -> I/O error; offsets damaged inside an otherwise complete file -> argument error
    path = tmp_path / "g.vdbgraf"; hp.save(path)
    raw = path.read_bytes()
    (tmp_path / "short").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(vdb.DBError) as ei:
        hp.load(tmp_path / "short")
    assert ei.value.code == 3
    with pytest.raises(vdb.DBError) as ei:
        hp.load(tmp_path / "does-not-exist")
    assert ei.value.code == 3
    n = len(g["ids"])
    layer0 = 112 + n * 8 + n * 4 + ((n + 7) // 8) * 8 + n * 12 * 4        # header, ids, levels, deleted (padded), vectors
    n_lists = int(np.frombuffer(raw[layer0:layer0 + 8], "<u8")[0])
    off0 = layer0 + 16 + ((n_lists * 4 + 7) // 8) * 8                      # counts, owners (padded)
    bad = bytearray(raw); bad[off0 + 8 * 3: off0 + 8 * 4] = np.uint64(10 ** 9).tobytes()
    (tmp_path / "offsets").write_bytes(bytes(bad))
    with pytest.raises(vdb.ArgumentError):
        hp.load(tmp_path / "offsets")
    huge = bytearray(raw); huge[48:56] = np.uint64(2 ** 40).tobytes()        # header field n
    (tmp_path / "huge").write_bytes(bytes(huge))
    with pytest.raises(vdb.ArgumentError):
        hp.load(tmp_path / "huge")
    got, gotn = hp.search_batch(q, 5)
    assert np.array_equal(refn, gotn) and np.array_equal(ref["id"], got["id"])


def _seg_payload(api, i):
    p = api.Payload()
    p.set("cat", api.PayloadValue.Int(i % 5)); p.set("w", api.PayloadValue.Float(0.25 * (i % 7)))
    if i % 3 == 0: p.set("name", api.PayloadValue.Str(["a", "bb", "ccc é"][i % 9 // 3]))
    if i % 4 == 0: p.set("on", api.PayloadValue.Bool(i % 8 == 0))
    if i % 10 == 0: p.set("tags", api.PayloadValue.ListStr(["x", "yy"][: 1 + i % 20 // 10]))
    if i % 25 == 0: p.set("nums", api.PayloadValue.ListInt([i, -i])); p.set("fs", api.PayloadValue.ListFloat([0.5 * i])); p.set("bs", api.PayloadValue.ListBool([True, False]))
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 2])
def test_a_saved_and_loaded_segment_answers_and_grows_like_the_original(ora, vdb, tmp_path, metric):
    """vdb_segment_save / vdb_segment_load: the loaded segment holds the same payloads, tombstones and next id, answers
    search_with_filter bit-identically, and further inserts build the graph the oracle's segment builds from the same
    history (same seed, STRICT arithmetic) — the random streams continue across the file."""
    n, extra, d = 260, 40, 10
    rng = np.random.default_rng(70 + metric)
    x = rng.uniform(-1, 1, size=(n + extra, d)).astype(np.float32)
    def fresh(api, seed=8, **kw): return api.Segment(api.HNSWIndex(metric, 8, 32, 16, d, seed=seed, **kw))
    so, sv = fresh(ora), fresh(vdb, arith=vdb.Arith.Strict)
    for i in range(n):
        a = so.insert(x[i], _seg_payload(ora, i) if i % 6 else None); b = sv.insert(x[i], _seg_payload(vdb, i) if i % 6 else None)
        assert a == b
    for i in (3, 50, 51, 200):
        so.delete(i + 1); sv.delete(i + 1)
    path = tmp_path / "seg.vdbseg"
    sv.save(path)
    # the file as the Python reader sees it
    cfg, g, seg = graph_file.read_segment(path)
    assert seg["next_id"] == n + 1 and seg["deleted"] == [4, 51, 52, 201] and len(seg["payloads"]) == sv.payload_count()
    assert seg["payloads"][2]["cat"] == ("Int", 1) and 1 not in seg["payloads"] and seg["payloads"][26]["nums"] == ("ListInt", [25, -25]) and seg["payloads"][16]["name"] == ("Str", "ccc é")
    assert cfg["dim"] == d and len(g["ids"]) == n
    # loaded into a segment created with another seed: everything comes from the file
    sl = fresh(vdb, seed=99, arith=vdb.Arith.Strict)
    sl.load(path)
    assert sl.payload_count() == sv.payload_count() and sl.is_deleted(51) and not sl.is_deleted(2) and sl.get_vector(4) is None
    assert sl.payload_index().query_exact("cat", vdb.PayloadValue.Int(2)) == sv.payload_index().query_exact("cat", vdb.PayloadValue.Int(2))
    V, F, Op = vdb.PayloadValue, vdb.Filter, vdb.ScalarComparisonOp
    VO, FO, OpO = ora.PayloadValue, ora.Filter, ora.ScalarComparisonOp
    flt = lambda A, B, C: [None, A.Match("cat", B.Int(3)), A.And([A.Compare("w", C.Gte, B.Float(0.5)), A.Not(A.Match("name", B.Str("bb")))]), A.Compare("tags", C.Eq, B.Str("yy"))]
    def same(a, b):
        assert [r.id for r in a] == [r.id for r in b] and [np.float32(r.sort_key).view(np.uint32) for r in a] == [np.float32(r.sort_key).view(np.uint32) for r in b]
    for fv, fo in zip(flt(F, V, Op), flt(FO, VO, OpO)):
        for qi in range(0, n, 23):
            for call in ("search_with_filter", "post_filter"):
                try: want = getattr(so, call)(x[qi], 5, fo)
                except ora.DBError as e: want = type(e).__name__
                try: got = getattr(sl, call)(x[qi], 5, fv)
                except vdb.DBError as e: got = type(e).__name__
                if isinstance(want, str) or isinstance(got, str): assert want == got
                else: same(want, got)
    # the history continues identically: ids, graph, results
    for i in range(n, n + extra):
        a = so.insert(x[i], _seg_payload(ora, i)); b = sl.insert(x[i], _seg_payload(vdb, i)); c = sv.insert(x[i], _seg_payload(vdb, i))
        assert a == b == c
    _same_graph(so.hnsw().export_graph(), sl.hnsw().export_graph())
    _same_graph(sv.hnsw().export_graph(), sl.hnsw().export_graph())
    # damaged files leave the segment as it was
    raw = path.read_bytes()
    (tmp_path / "cut").write_bytes(raw[:-20])
    with pytest.raises(vdb.DBError):
        sl.load(tmp_path / "cut")
    at = raw.index(b"VDBSEGM1")
    (tmp_path / "magic").write_bytes(raw[:at] + b"VDBSEGMX" + raw[at + 8:])
    with pytest.raises(vdb.ArgumentError):
        sl.load(tmp_path / "magic")
    bad = bytearray(raw); bad[at + 32: at + 40] = np.uint64(123456).tobytes()        # first deleted id: not a tombstone of the index
    (tmp_path / "del").write_bytes(bytes(bad))
    with pytest.raises(vdb.ArgumentError):
        sl.load(tmp_path / "del")
    assert sl.hnsw().len() == n + extra and sl.payload_count() == sv.payload_count()
    same(sl.search(x[7], 5), sv.search(x[7], 5))
    # an index file loads the index part of a segment file
    hi = vdb.HNSWIndex(metric, 8, 32, 16, d, arith=vdb.Arith.Strict); hi.load(path)
    assert hi.len() == n
