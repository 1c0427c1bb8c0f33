"""Restates /root/reference/tests/hnsw.rs (14 tests) against the oracle (CPU) and the HIP library (GPU)."""
import numpy as np
import pytest


def generate_points(n, dim, spread, seed):                      # hnsw.rs:12-26 (seeded here)
    rng = np.random.default_rng(seed)
    return rng.uniform(-spread, spread, size=(n, dim)).astype(np.float32)


def test_all_metrics_consistency(be):                           # hnsw.rs:29-89
    M = be.DistanceMetric
    for k, metric in enumerate([M.Euclidean, M.Cosine, M.Dot]):
        hnsw = be.HNSWIndex(metric, 16, 64, 16, 4, seed=11 + k)
        points = generate_points(50, 4, 10.0, seed=100 + k)
        for i, v in enumerate(points):
            hnsw.insert(i, v)
        query = points[0]
        results = hnsw.search(query, 10)
        assert results
        best = results[0]
        if metric == M.Dot:
            max_dot = max(be.score(query, v, M.Dot) for v in points)
            assert abs(best.raw_score - max_dot) < 1e-5
        else:
            assert be.score(query, points[best.id], metric) <= 1e-4


def test_hnsw_robust_score_metrics(be):                         # hnsw.rs:92-122
    M = be.DistanceMetric
    vectors = [[float(x), 0.0, 0.0] for x in (1, 2, 3, 4, 5)]
    he = be.HNSWIndex(M.Euclidean, 16, 50, 16, 3)
    for i, v in enumerate(vectors):
        he.insert(i, v)
    assert he.search([1.1, 0.0, 0.0], 1)[0].id == 0
    hc = be.HNSWIndex(M.Cosine, 16, 50, 16, 3)
    for i, v in enumerate(vectors):
        hc.insert(i, v)
    assert hc.search([4.9, 0.0, 0.0], 1)[0].id in (0, 1, 2, 3, 4)


def test_large_insertion_and_ranking_accuracy(be):              # hnsw.rs:125-164
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 32, 64, 16, 3)
    for i in range(300):
        hnsw.insert(i, [float(i)] * 3)
    results = hnsw.search([0.0, 0.0, 0.0], 5)
    assert results[0].id == 0
    assert len(results) == 5
    for a, b in zip(results, results[1:]):
        assert a.raw_score <= b.raw_score


def test_dot_product_prefers_larger_magnitudes(be):             # hnsw.rs:166-192
    hnsw = be.HNSWIndex(be.DistanceMetric.Dot, 16, 50, 16, 2)
    hnsw.insert(1, [1.0, 1.0]); hnsw.insert(2, [10.0, 10.0]); hnsw.insert(3, [-1.0, -1.0])
    assert hnsw.search([1.0, 1.0], 3)[0].id == 2


def test_cosine_distance_with_opposite_vectors(be):             # hnsw.rs:194-218
    hnsw = be.HNSWIndex(be.DistanceMetric.Cosine, 16, 50, 16, 3)
    hnsw.insert(1, [1.0, 0.0, 0.0]); hnsw.insert(2, [-1.0, 0.0, 0.0])
    r = hnsw.search([1.0, 0.0, 0.0], 2)
    assert r[0].id == 1 and r[1].id == 2


def test_idempotent_insert_and_query(be):                       # hnsw.rs:220-244
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 2)
    hnsw.insert(1, [3.0, 4.0]); hnsw.insert(1, [3.0, 4.0])
    r = hnsw.search([3.0, 4.0], 1)
    assert len(r) == 1 and r[0].id == 1
    assert hnsw.len() == 1


def test_empty_index_search_returns_empty(be):                  # hnsw.rs:246-259
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 2)
    assert hnsw.search([0.0, 0.0], 10) == []


def test_dimension_mismatch_is_handled(be):                     # hnsw.rs:261-299
    hnsw = be.HNSWIndex(be.DistanceMetric.Cosine, 16, 50, 16, 5)
    hnsw.insert(1, [1.0] * 5)
    bad = [1.0, 2.0]
    with pytest.raises(be.VectorLengthMismatch) as e:
        hnsw.insert(2, bad)
    assert (e.value.expected, e.value.actual) == (5, 2)
    with pytest.raises(be.VectorLengthMismatch) as e:
        hnsw.search(bad, 1)
    assert (e.value.expected, e.value.actual) == (5, 2)


def test_search_k_greater_than_total_points(be):                # hnsw.rs:302-310
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 2)
    hnsw.insert(1, [1.0, 2.0]); hnsw.insert(2, [2.0, 3.0])
    assert len(hnsw.search([1.5, 2.5], 10)) == 2


def test_single_insertion_exact_retrieval(be):                  # hnsw.rs:312-322
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 2)
    hnsw.insert(42, [3.14, 2.71])
    r = hnsw.search([3.14, 2.71], 1)
    assert len(r) == 1 and r[0].id == 42 and r[0].raw_score <= 1e-6


def test_insertion_order_independence(be):                      # hnsw.rs:324-336
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 2)
    hnsw.insert(100, [5.0, 5.0]); hnsw.insert(200, [1.0, 1.0]); hnsw.insert(300, [3.0, 3.0])
    assert hnsw.search([2.9, 2.9], 1)[0].id == 300


def test_dense_cloud_retrieval_accuracy(be):                    # hnsw.rs:339-353
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 64, 16, 3)
    for i in range(500):
        off = np.float32(i) * np.float32(0.01)
        hnsw.insert(i, [off, off, off])
    assert hnsw.search([0.0, 0.0, 0.0], 5)[0].id == 0


def test_high_dimensional_accuracy(be):                         # hnsw.rs:355-373
    dim = 64
    hnsw = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 64, 16, dim)
    hnsw.insert(1, [1.0] * dim); hnsw.insert(2, [100.0] * dim)
    assert hnsw.search([1.0] * dim, 1)[0].id == 1


def test_stable_search_results(be):                             # hnsw.rs:376-390
    hnsw = be.HNSWIndex(be.DistanceMetric.Cosine, 16, 64, 16, 3)
    for i in range(100):
        hnsw.insert(i, [float(i), 1.0, 0.0])
    q = [50.0, 1.0, 0.0]
    assert hnsw.search(q, 5) == hnsw.search(q, 5)


def test_accessors(be):                                         # hnsw.rs:691-750
    hnsw = be.HNSWIndex(be.DistanceMetric.Cosine, 8, 32, 4, 3)
    assert (hnsw.m(), hnsw.ef(), hnsw.max_level_cap(), hnsw.dim()) == (8, 32, 4, 3)
    assert hnsw.metric() == be.DistanceMetric.Cosine
    assert hnsw.get_entry_point() is None
    hnsw.insert(7, [3.0, 0.0, 4.0])
    assert hnsw.get_entry_point() == 7 and hnsw.contains(7) and not hnsw.contains(8)
    np.testing.assert_allclose(hnsw.get_vector(7), [0.6, 0.0, 0.8], rtol=1e-6)   # stored normalised, hnsw.rs:138
    assert hnsw.layer_neighbors(0, 7) == [7]                                      # self link, hnsw.rs:143-148
    assert hnsw.layer_neighbors(0, 99) is None
    assert hnsw.normalize_score(2.0) == 2.0
    hnsw.mark_deleted(7)
    assert hnsw.get_vector(7) is None and hnsw.get_entry_point() is None
