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
