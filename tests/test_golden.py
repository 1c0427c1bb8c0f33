"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).

Run against the CPU checker (pins it against drift) and, on a GPU box, against the HIP library in
strict arithmetic, where every id, raw score and sort key must match bit for bit: once by
importing the stored graph, once by rebuilding it with the literal insert sequence.
"""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX_FILES = sorted(glob.glob(os.path.join(GOLDEN, "index_*.npz")))


def _graph(d):
    layers = [(d[f"g_owner{l}"], d[f"g_off{l}"], d[f"g_nb{l}"]) for l in range(int(d["g_layers"][0]))]
    e = int(d["g_entry"][0])
    return dict(ids=d["g_ids"], levels=d["g_levels"], vectors=d["g_vectors"], deleted=d["g_deleted"], layers=layers,
                entry=None if e < 0 else e, current_max_level=int(d["g_max_level"][0]))


def _same_lists(out, cnt, exp, exp_n):
    assert np.array_equal(np.asarray(cnt, np.int64), np.asarray(exp_n, np.int64))
    for q in range(len(exp)):
        n = int(exp_n[q])
        assert np.array_equal(out["id"][q][:n], exp["id"][q][:n]), f"query {q}"
        assert out["raw_score"][q][:n].tobytes() == exp["raw_score"][q][:n].tobytes(), f"query {q}"
        assert out["sort_key"][q][:n].tobytes() == exp["sort_key"][q][:n].tobytes(), f"query {q}"


def _same_graph(a, b):
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["levels"], b["levels"])
    assert a["vectors"].tobytes() == b["vectors"].tobytes()
    assert np.array_equal(a["deleted"], b["deleted"])
    assert a["entry"] == b["entry"] and a["current_max_level"] == b["current_max_level"]
    assert len(a["layers"]) == len(b["layers"])
    for (ow, of, nb), (ow2, of2, nb2) in zip(a["layers"], b["layers"]):
        assert np.array_equal(ow, ow2) and np.array_equal(of, of2) and np.array_equal(nb, nb2)


def test_fixture_files_present():
    assert len(INDEX_FILES) == 4 and os.path.exists(os.path.join(GOLDEN, "segment_l2_32.npz"))


@pytest.mark.parametrize("path", INDEX_FILES, ids=[os.path.basename(p)[6:-4] for p in INDEX_FILES])
@pytest.mark.parametrize("how", ["import", "rebuild"])
def test_index_golden(be, path, how):
    d = np.load(path)
    m, ef, cap, dim, top_k, seed = (int(x) for x in d["params"])
    metric = be.DistanceMetric(int(d["metric"][0]))
    idx = be.HNSWIndex(metric, m, ef, cap, dim, seed=seed, arith=be.Arith.Strict)
    if how == "import":
        idx.import_graph(_graph(d))
    else:
        base = d["base"] if "base" in d.files else d["g_vectors"]
        for i, pid in enumerate(d["ids"]):
            idx.insert(int(pid), base[i])
        for s in d["deleted"]:
            idx.mark_deleted(int(d["ids"][int(s)]))
        _same_graph(idx.export_graph(), _graph(d))
    for e in d["ef_list"]:
        out, cnt, st = idx.search_batch(d["queries"], top_k, ef_search=int(e), with_stats=True)
        _same_lists(out, cnt, d[f"ann{e}"], d[f"ann{e}_n"])
        assert np.array_equal(st["expanded"], d[f"ann{e}_expanded"])
    out, cnt = idx.exact_knn_batch(d["queries"], top_k)
    _same_lists(out, cnt, d["exact"], d["exact_n"])
    gone = set(int(d["ids"][int(s)]) for s in d["deleted"])
    assert not gone & set(int(x) for x in out["id"].ravel())


def test_segment_golden(be):
    d = np.load(os.path.join(GOLDEN, "segment_l2_32.npz"))
    m, ef, cap, dim, top_k, seed = (int(x) for x in d["params"])
    base, queries = d["g_vectors"], d["queries"]
    seg = be.Segment(be.HNSWIndex(be.DistanceMetric.Euclidean, m, ef, cap, dim, seed=seed, arith=be.Arith.Strict))
    PV, F, Op = be.PayloadValue, be.Filter, be.ScalarComparisonOp
    for i in range(len(base)):
        p = be.Payload(); p.set("cat", PV.Int(i % 7)); p.set("tier", PV.Int(i % 3))
        if i % 5 == 0:
            p.set("label", PV.Str("x"))
        assert seg.insert(base[i], p) == i + 1
    _same_graph(seg.hnsw().export_graph(), _graph(d))
    filters = {"match": F.Match("cat", PV.Int(3)),
               "and": F.And([F.Match("tier", PV.Int(1)), F.Compare("cat", Op.Lt, PV.Int(4))]),
               "not": F.Not(F.Match("tier", PV.Int(0)))}

    def check(tag):
        calls = [(f"{tag}_search", lambda q: seg.search(q, top_k))]
        for fname, f in filters.items():
            calls.append((f"{tag}_search_with_filter_{fname}", lambda q, f=f: seg.search_with_filter(q, top_k, f)))
            calls.append((f"{tag}_post_filter_{fname}", lambda q, f=f: seg.post_filter(q, top_k, f)))
        for key, call in calls:
            exp, exp_n = d[key], d[key + "_n"]
            for q in range(len(queries)):
                got = call(queries[q])
                assert len(got) == int(exp_n[q]), (key, q)
                assert [r.id for r in got] == [int(x) for x in exp["id"][q][: len(got)]], (key, q)
                assert np.array([r.raw_score for r in got], np.float32).tobytes() == exp["raw_score"][q][: len(got)].tobytes(), (key, q)

    check("fresh")
    for i in d["deletes"]:
        seg.delete(int(i))
    check("deleted")
