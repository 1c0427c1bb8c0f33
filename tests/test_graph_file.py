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
    assert cfg["seed"] == 2 and cfg["dim"] == 20
    hq = vdb.HNSWIndex(metric, 8, 32, 16, 20, seed=2, arith=vdb.Arith.Strict)
    hq.load(p2)
    again, againn = hq.search_batch(q, 10)
    assert np.array_equal(got["id"], again["id"]) and np.array_equal(gotn, againn)
    # a built (not imported) library index keeps its random-stream positions across save/load: the next insert draws the same level
    hb, xb = _build(vdb, metric, 200, 20, 33, arith=vdb.Arith.Strict)
    p3 = tmp_path / "built.vdbgraf"; hb.save(p3)
    hc = vdb.HNSWIndex(metric, 8, 32, 16, 20, seed=33, arith=vdb.Arith.Strict); hc.load(p3)
    extra = np.random.default_rng(1).uniform(-1, 1, size=(40, 20)).astype(np.float32)
    for i in range(40):
        hb.insert(10_000 + i, extra[i]); hc.insert(10_000 + i, extra[i])
    _same_graph(hb.export_graph(), hc.export_graph())
    with pytest.raises(vdb.DBError):
        vdb.HNSWIndex(metric, 8, 32, 16, 21).load(p3)
