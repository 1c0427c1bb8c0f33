"""Edge cases of the batched entry points (extensions of HNSWIndex::search, hnsw.rs:513-554, to many queries): empty
batches, top_k against ef and against the number of points, tiny indexes, the smallest ef, the largest supported
dimension.  Expected values come from the oracle's single-query search (same graph through import) or from the
reference's own rules: top_k beyond ef is truncated to ef (hnsw.rs:549-551), k beyond the number of points returns all
of them (tests/hnsw.rs:302-310), an empty index returns nothing (hnsw.rs:515-518), a query of the wrong length is
VectorLengthMismatch (hnsw.rs:519-524).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pair(vdb, ora, metric, n, d, seed, m=8, ef=32, arith=None, **kw):
    """The same graph in the oracle and in the library (built by the oracle, imported)."""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, d)).astype(np.float32)
    ho = ora.HNSWIndex(metric, m, ef, 8, d, seed=seed)
    for i in range(n):
        ho.insert(i + 1, x[i])
    hp = vdb.HNSWIndex(metric, m, ef, 8, d, seed=seed, arith=arith if arith is not None else vdb.Arith.Strict, **kw)
    hp.import_graph(ho.export_graph())
    return ho, hp, x, rng


def test_empty_batch_and_empty_index(vdb):
    h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 8, 32, 8, 16)
    out, cnt = h.search_batch(np.zeros((5, 16), np.float32), 3)
    assert (cnt == 0).all()                                           # hnsw.rs:515-518
    out, cnt = h.exact_knn_batch(np.zeros((5, 16), np.float32), 3)
    assert (cnt == 0).all()
    h.insert(7, np.ones(16, np.float32))
    out, cnt = h.search_batch(np.zeros((0, 16), np.float32), 3)       # no queries: nothing to do, no error
    assert out.shape == (0, 3) and cnt.shape == (0,)
    out, cnt = h.exact_knn_batch(np.zeros((0, 16), np.float32), 3)
    assert out.shape == (0, 3)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_top_k_against_ef_and_against_the_number_of_points(vdb, ora, metric):
    ho, hp, x, rng = _pair(vdb, ora, metric, 200, 12, 5 + metric, ef=16)
    q = rng.normal(size=(9, 12)).astype(np.float32)
    # top_k beyond ef: ef results (hnsw.rs:549-551), the oracle's
    out, cnt = hp.search_batch(q, 40)
    for i in range(len(q)):
        ref = ho.search(q[i], 40)
        assert cnt[i] == len(ref) <= 16
        assert [r.id for r in ref] == out["id"][i][: cnt[i]].tolist()
        assert np.array_equal(np.array([r.sort_key for r in ref], np.float32).view(np.uint32), out["sort_key"][i][: cnt[i]].view(np.uint32))
    # ef_search = 1: one result, the oracle's
    out, cnt = hp.search_batch(q, 5, ef_search=1)
    ref, refn = ho.search_batch(q, 5, ef_search=1)
    assert np.array_equal(cnt, refn) and (cnt == 1).all() and np.array_equal(out["id"][:, 0], ref["id"][:, 0])
    # exact k-NN with k beyond the number of points: all of them, ascending (tests/hnsw.rs:302-310)
    out, cnt = hp.exact_knn_batch(q, 256)
    ref, refn = ho.exact_knn_batch(q, 256)
    assert (cnt == 200).all() and np.array_equal(cnt, refn)
    assert np.array_equal(out["id"][:, :200], ref["id"][:, :200])
    with pytest.raises(vdb.DBError):                                   # the exact search takes top_k up to 256 and says so
        hp.exact_knn_batch(q, 400)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_tiny_indexes(vdb, ora, n):
    for arith in (vdb.Arith.Strict, vdb.Arith.Fast):
        ho, hp, x, rng = _pair(vdb, ora, 2, n, 7, 40 + n, arith=arith)
        q = np.concatenate([x, rng.normal(size=(4, 7)).astype(np.float32)])
        out, cnt = hp.search_batch(q, 5)
        ref, refn = ho.search_batch(q, 5)
        assert np.array_equal(cnt, refn) and (cnt == n).all()
        assert np.array_equal(out["id"][:, :n], ref["id"][:, :n])
        ex, exn = hp.exact_knn_batch(q, 5)
        assert (exn == n).all() and (ex["sort_key"][:n, 0] == 0).all()          # a stored row first, at distance 0


def test_wrong_query_length_and_the_largest_dimension(vdb, ora):
    h = vdb.HNSWIndex(vdb.DistanceMetric.Dot, 8, 32, 8, 16)
    h.insert(1, np.ones(16, np.float32))
    with pytest.raises(vdb.DBError) as e:
        h.search_batch(np.zeros((2, 15), np.float32), 3)
    assert "mismatch" in str(e.value).lower()                          # hnsw.rs:519-524
    # 2048 floats per row is the longest the FAST kernels take; the oracle's answer on the same graph (one node per
    # step: the reference's order of expansion)
    ho, hp, x, rng = _pair(vdb, ora, 1, 200, 2048, 77, arith=vdb.Arith.Fast, beam_width=1)
    q = rng.normal(size=(6, 2048)).astype(np.float32)
    out, cnt = hp.search_batch(q, 5)
    ref, refn = ho.search_batch(q, 5)
    assert np.array_equal(cnt, refn)
    assert np.array_equal(out["id"], ref["id"])
    assert np.allclose(out["raw_score"], ref["raw_score"], rtol=1e-5, atol=1e-5 * float(np.abs(ref["raw_score"]).max()))
    ex, exn = hp.exact_knn_batch(q, 5)
    rex, rexn = ho.exact_knn_batch(q, 5)
    assert np.array_equal(ex["id"], rex["id"])


def test_segment_batches_with_filters_that_pass_nothing_everything_and_after_deleting_all(vdb):
    """Segment::search_with_filter in batches (segment.rs:121-147; Masked mode): a filter no payload satisfies returns
    nothing, one that every payload satisfies returns k passing rows in ascending order — also for k beyond the 64
    rows the register-resident passing list holds —, and a segment whose points are all deleted is SearchError
    (segment.rs:104-107, 126-130)."""
    n, d = 1500, 24
    rng = np.random.default_rng(91)
    x = rng.normal(size=(n, d)).astype(np.float32)
    seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 8, 64, 16, d, seed=3, arith=vdb.Arith.Fast, filter_mode=vdb.FilterMode.Masked))
    pays = []
    for i in range(n):
        p = vdb.Payload(); p.set("all", vdb.PayloadValue.Int(1)); p.set("odd", vdb.PayloadValue.Int(i % 2)); pays.append(p)
    ids = seg.insert_batch(x, pays)
    q = rng.normal(size=(33, d)).astype(np.float32)
    out, cnt = seg.search_batch(q, 10, filter=vdb.Filter.Match("all", vdb.PayloadValue.Int(7)))
    assert (cnt == 0).all()
    for k in (10, 100):
        out, cnt = seg.search_batch(q, k, ef_search=128, filter=vdb.Filter.Match("all", vdb.PayloadValue.Int(1)))
        assert (cnt == k).all() and (np.diff(out["sort_key"], axis=1) >= 0).all()
        out, cnt = seg.search_batch(q, k, ef_search=128, filter=vdb.Filter.Match("odd", vdb.PayloadValue.Int(1)))
        assert (cnt == k).all() and (np.diff(out["sort_key"], axis=1) >= 0).all()
        odd = {int(ids[i]) for i in range(n) if i % 2 == 1}
        assert all(int(v) in odd for v in out["id"].reshape(-1))
        # distances are the rows' own (float64 recomputation)
        slot = {int(ids[i]): i for i in range(n)}
        for qi in (0, 17):
            rows = x[[slot[int(v)] for v in out["id"][qi]]]
            ref = np.sqrt(((rows.astype(np.float64) - q[qi].astype(np.float64)) ** 2).sum(axis=1))
            assert np.allclose(out["sort_key"][qi], ref, rtol=1e-5)
    for i in ids:
        seg.delete(int(i))
    with pytest.raises(vdb.DBError) as e:
        seg.search_batch(q, 10)
    assert "No active points" in str(e.value)
