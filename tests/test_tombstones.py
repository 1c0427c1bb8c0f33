"""Tombstones recorded before their point exists (hnsw.rs:111-117: the deleted set takes any id)."""
import numpy as np


def test_an_id_marked_deleted_before_its_insert_is_born_deleted(be):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(60, 6)).astype(np.float32)
    h = be.HNSWIndex(be.DistanceMetric.Euclidean, 8, 32, 16, 6, seed=2)
    h.mark_deleted(1000 + 17)                      # unknown for now
    for i in range(60):
        h.insert(1000 + i, x[i])
    assert h.contains(1017) and h.get_vector(1017) is None          # hnsw.rs:727-734 hides deleted points
    out = h.search(x[17], 5)
    assert 1017 not in [r.id for r in out] and len(out) == 5
    # an ordinary delete behaves as before
    h.mark_deleted(1003)
    assert 1003 not in [r.id for r in h.search(x[3], 5)]
