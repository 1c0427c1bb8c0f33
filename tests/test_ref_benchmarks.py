"""Restates the workloads of /root/reference/tests/benchmarks.rs (no timing assertions there either)."""
import numpy as np
import pytest

f32 = np.float32


def generate_vector(i):                                            # benchmarks.rs:14-20
    return np.array([np.sin(f32(i)) * f32(5.0), np.cos(f32(i * 3)) * f32(3.0), np.sqrt(f32(i % 7))], dtype=f32)


def generate_payload(be, i):                                        # benchmarks.rs:22-40
    PV = be.PayloadValue
    p = be.Payload()
    p.set("index", PV.Int(i))
    p.set("animal", PV.Str(["dog", "cat", "bird", "fish"][i % 4]))
    p.set("age", PV.Int(i % 8 + 1))
    p.set("score", PV.Float(60.0 + (i % 40)))
    p.set("tags", PV.ListStr(["cheap", "small"] if i % 2 == 0 else ["expensive", "large"]))
    p.set("active", PV.Bool(i % 3 == 0))
    return p


@pytest.mark.parametrize("metric_i,size", [(2, 100), (0, 1000), (1, 500)])   # benchmarks.rs:117-145
def test_bench_all(be, metric_i, size):
    M, PV, F, Op = be.DistanceMetric, be.PayloadValue, be.Filter, be.ScalarComparisonOp
    metric = [M.Cosine, M.Dot, M.Euclidean][metric_i]
    seg = be.Segment(be.HNSWIndex(metric, 16, 50, 16, 3, seed=size))
    for i in range(size):
        seg.insert(generate_vector(i), generate_payload(be, i))
    q = [1.0, 0.0, 0.0]
    assert len(seg.search(q, 10)) == 10
    flt = F.And([F.Match("animal", PV.Str("dog")), F.Compare("age", Op.Gte, PV.Int(6)),
                 F.Compare("score", Op.Lt, PV.Float(90.0))])
    for r in seg.post_filter(q, 10, flt):
        p = seg.get_payload(r.id)
        assert p.get("animal") == PV.Str("dog")
    tag = F.Compare("tags", Op.Eq, PV.Str("cheap"))
    for r in seg.post_filter(q, 10, tag):
        assert (r.id - 1) % 2 == 0
    for i in range(0, size, 7):                                     # benchmarks.rs:100-114
        seg.delete(i + 1)
    for i in range(size):
        seg.delete(i + 1)
    for i in range(size):
        assert seg.get_vector(i + 1) is None
