"""Regenerates tests/golden/*.npz with the CPU oracle:  python tests/golden/make_golden.py

The reference ships no golden files (its tests are property checks), and it cannot be run in this
build environment (no Rust toolchain), so these vectors are produced by the oracle restatement and
pin (i) the oracle against silent drift and (ii) the HIP library on a GPU box without needing
anything but the fixture.  Every case stores its inputs (vectors, ids, queries, parameters), the
graph the literal insert sequence produces, and the expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

INDEX_CASES = [
    # name, metric, dim, n, m, ef, cap, nq, top_k, ef_search list, deleted ids, data kind
    ("cos3", "Cosine", 3, 600, 8, 32, 8, 32, 5, [0, 64], [], "uniform"),
    ("dot32", "Dot", 32, 1500, 16, 64, 16, 48, 10, [0, 24], list(range(7, 1500, 37)), "uniform"),
    ("l2_96", "Euclidean", 96, 1500, 16, 64, 16, 48, 10, [16, 0, 200], [], "uniform"),
    ("l2_128c", "Euclidean", 128, 2000, 16, 64, 16, 64, 10, [0, 128], [], "clustered"),
]


def make_data(kind, n, dim, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(n, dim)).astype(np.float32)
    centres = rng.uniform(0, 128, size=(32, dim)).astype(np.float32)
    pick = rng.integers(0, len(centres), size=n)
    return np.maximum(centres[pick] + rng.normal(0, 12, size=(n, dim)).astype(np.float32), 0).astype(np.float32)


def flatten_graph(g):
    out = dict(g_ids=g["ids"], g_levels=g["levels"], g_vectors=g["vectors"], g_deleted=g["deleted"],
               g_entry=np.array([-1 if g["entry"] is None else g["entry"]], np.int64),
               g_max_level=np.array([g["current_max_level"]], np.int64), g_layers=np.array([len(g["layers"])], np.int64))
    for l, (ow, of, nb) in enumerate(g["layers"]):
        out[f"g_owner{l}"], out[f"g_off{l}"], out[f"g_nb{l}"] = ow, of, nb
    return out


def index_case(ora, name, metric, dim, n, m, ef, cap, nq, top_k, ef_list, deleted, kind):
    seed = sum(name.encode()) + 1000
    base = make_data(kind, n, dim, seed)
    queries = make_data(kind, nq, dim, seed + 1)
    queries[0] = base[n // 2]                                # a query that equals a stored row
    ids = (np.arange(n, dtype=np.uint64) * 3 + 11)           # ids are not slots
    idx = ora.HNSWIndex(getattr(ora.DistanceMetric, metric), m, ef, cap, dim, seed=seed)
    for i in range(n):
        idx.insert(int(ids[i]), base[i])
    for d in deleted:
        idx.mark_deleted(int(ids[d]))
    d = dict(metric=np.array([int(getattr(ora.DistanceMetric, metric))]), params=np.array([m, ef, cap, dim, top_k, seed], np.int64),
             base=base, ids=ids, queries=queries, deleted=np.array(deleted, np.int64), ef_list=np.array(ef_list, np.int64))
    d.update(flatten_graph(idx.export_graph()))
    for e in ef_list:
        out, cnt, st = idx.search_batch(queries, top_k, ef_search=e, with_stats=True)
        d[f"ann{e}"], d[f"ann{e}_n"], d[f"ann{e}_expanded"] = out, cnt, st["expanded"]
    d["exact"], d["exact_n"] = idx.exact_knn_batch(queries, top_k)
    if metric != "Cosine":                                   # stored form == input: keep one copy (g_vectors)
        assert d["g_vectors"].tobytes() == base.tobytes()
        del d["base"]
    return d


def segment_case(ora):
    """Segment with payloads: filter-aware edges at insert, in-place filtered search, post_filter, deletes."""
    n, dim, nq, top_k, seed = 1200, 32, 32, 8, 4242
    base = make_data("uniform", n, dim, seed); queries = make_data("uniform", nq, dim, seed + 1)
    seg = ora.Segment(ora.HNSWIndex(ora.DistanceMetric.Euclidean, 8, 48, 12, dim, seed=seed))
    PV, F, Op = ora.PayloadValue, ora.Filter, ora.ScalarComparisonOp
    for i in range(n):
        p = ora.Payload(); p.set("cat", PV.Int(i % 7)); p.set("tier", PV.Int(i % 3))
        if i % 5 == 0:
            p.set("label", PV.Str("x"))                      # "label" missing elsewhere: Compare on it errs
        assert seg.insert(base[i], p) == i + 1
    filters = {"match": F.Match("cat", PV.Int(3)),
               "and": F.And([F.Match("tier", PV.Int(1)), F.Compare("cat", Op.Lt, PV.Int(4))]),
               "not": F.Not(F.Match("tier", PV.Int(0)))}
    d = dict(params=np.array([8, 48, 12, dim, top_k, seed], np.int64), base=base, queries=queries)
    d.update(flatten_graph(seg.hnsw().export_graph()))
    assert d["g_vectors"].tobytes() == base.tobytes()
    del d["base"]

    def pack(rows):
        out = np.zeros(top_k, ora.SCORED_DTYPE)
        for j, r in enumerate(rows):
            out[j] = (r.id, r.raw_score, r.sort_key)
        return out, len(rows)

    def record(tag):
        for fname, f in filters.items():
            for kind in ("search_with_filter", "post_filter"):
                res = [pack(getattr(seg, kind)(queries[q], top_k, f)) for q in range(nq)]
                d[f"{tag}_{kind}_{fname}"] = np.stack([r[0] for r in res])
                d[f"{tag}_{kind}_{fname}_n"] = np.array([r[1] for r in res], np.int64)
        res = [pack(seg.search(queries[q], top_k)) for q in range(nq)]
        d[f"{tag}_search"] = np.stack([r[0] for r in res]); d[f"{tag}_search_n"] = np.array([r[1] for r in res], np.int64)

    record("fresh")
    dels = list(range(5, n + 1, 9))                          # 133 deletes of 1200: below the 25 % purge bar
    for i in dels:
        seg.delete(i)
    d["deletes"] = np.array(dels, np.int64)
    record("deleted")
    return d


def main():
    from oracle import pyoracle
    ora = pyoracle.load()
    for case in INDEX_CASES:
        np.savez_compressed(os.path.join(HERE, f"index_{case[0]}.npz"), **index_case(ora, *case))
        print("wrote", case[0])
    np.savez_compressed(os.path.join(HERE, "segment_l2_32.npz"), **segment_case(ora))
    print("wrote segment_l2_32")


if __name__ == "__main__":
    main()
