"""Restates /root/reference/tests/segment.rs (4 tests) against the oracle (CPU) and the HIP library (GPU)."""
import numpy as np
import pytest

f32 = np.float32


def metrics(be):
    M = be.DistanceMetric
    return [M.Euclidean, M.Cosine, M.Dot]


@pytest.mark.parametrize("mi", [0, 1, 2])
def test_large_scale_insert_and_search_all_metrics(be, mi):     # segment.rs:13-92
    metric = metrics(be)[mi]
    n = 10_000 if be.backend == "oracle" else 2_000             # single inserts are GPU round trips
    segment = be.Segment(be.HNSWIndex(metric, 16, 50, 16, 3, seed=5 + mi))
    vectors = []
    for i in range(n):
        vec = np.array([np.sin(f32(i)) * f32(5.0), np.cos(f32(i * 3)) * f32(3.0), np.sqrt(f32(i % 7))], dtype=f32)
        p = be.Payload()
        p.set("index", be.PayloadValue.Int(i))
        vectors.append((segment.insert(vec, p), vec))
    for expected_id, query in vectors[:10]:
        noisy = query + f32(0.001)
        results = segment.search(noisy, 5)
        if metric == be.DistanceMetric.Dot:
            best_id, best_dot = None, -np.inf
            for pid, v in vectors:
                d = f32(0)
                for a, b in zip(v, noisy):
                    d = f32(d + f32(a * b))
                if d > best_dot:
                    best_dot, best_id = d, pid
            assert results[0].id == best_id
        else:
            assert any(r.id == expected_id for r in results)


@pytest.mark.parametrize("mi", [0, 1, 2])
def test_large_scale_filtered_queries_all_metrics(be, mi):      # segment.rs:94-145
    metric = metrics(be)[mi]
    PV, F, Op = be.PayloadValue, be.Filter, be.ScalarComparisonOp
    segment = be.Segment(be.HNSWIndex(metric, 16, 50, 16, 3, seed=9))
    for i in range(1000):
        p = be.Payload()
        p.set("animal", PV.Str(["dog", "cat", "bird", "fish"][i % 4]))
        p.set("age", PV.Int(i % 8 + 1))
        p.set("score", PV.Float(60.0 + (i % 40)))
        e = np.exp(f32(i % 5))
        vec = [np.log1p(f32(i % 3)), e - np.trunc(e), np.power(f32(i % 7), f32(1.5))]
        segment.insert(vec, p)
    flt = F.And([F.Match("animal", PV.Str("dog")), F.Compare("age", Op.Gte, PV.Int(6)),
                 F.Compare("score", Op.Lt, PV.Float(90.0))])
    results = segment.post_filter([1.0, 0.0, 0.0], 15, flt)
    for r in results:
        p = segment.get_payload(r.id)
        assert p.get("animal") == PV.Str("dog")
        assert p.get("age").compare_scalar(Op.Gte, PV.Int(6)) is True
        assert p.get("score").compare_scalar(Op.Lt, PV.Float(90.0)) is True


@pytest.mark.parametrize("mi", [0, 1, 2])
def test_list_filters_with_larger_pool_all_metrics(be, mi):     # segment.rs:147-186
    metric = metrics(be)[mi]
    PV, F, Op = be.PayloadValue, be.Filter, be.ScalarComparisonOp
    segment = be.Segment(be.HNSWIndex(metric, 16, 50, 16, 3, seed=3))
    for i in range(1000):
        p = be.Payload()
        p.set("tags", PV.ListStr(["cheap", "small"] if i % 2 == 0 else ["expensive", "large"]))
        p.set("active", PV.Bool(i % 3 == 0))
        t = np.tan(f32(i))
        segment.insert([np.cos(f32(i)), np.sin(f32(i)), t - np.trunc(t)], p)
    flt = F.Compare("tags", Op.Eq, PV.Str("cheap"))
    results = segment.post_filter([0.0, 1.0, 0.0], 10, flt)
    assert len(results) >= 1
    for r in results:
        tags = segment.get_payload(r.id).get("tags")
        assert tags.evaluate_list_query(be.ListQueryKind.Contains, value=PV.Str("cheap")) is True


@pytest.mark.parametrize("mi", [0, 1, 2])
def test_deletion_and_purge_with_large_set_all_metrics(be, mi):  # segment.rs:188-223
    metric = metrics(be)[mi]
    segment = be.Segment(be.HNSWIndex(metric, 16, 50, 16, 3, seed=1))
    ids = []
    for i in range(200):
        p = be.Payload()
        p.set("idx", be.PayloadValue.Int(i))
        ids.append(segment.insert([float(i), 0.0, 0.0], p))
    assert ids == list(range(1, 201))                             # segment.rs:27,33,59
    for i in range(0, 200, 7):
        segment.delete(ids[i])
    for r in segment.search([10.0, 0.0, 0.0], 30):
        assert not segment.is_deleted(r.id)
    for pid in ids:
        segment.delete(pid)
    for pid in ids:
        assert segment.get_vector(pid) is None
    with pytest.raises(be.SearchError, match="No active points available to search."):
        segment.search([10.0, 0.0, 0.0], 3)                       # segment.rs:104-107
