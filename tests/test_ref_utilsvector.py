"""Restates /root/reference/tests/utilsvector.rs against the oracle (CPU) and the HIP library (GPU)."""
import pytest


def test_distance_vector_length_mismatch_panics(be):          # utilsvector.rs:7-13
    with pytest.raises(be.ReferencePanic, match="Vectors must be the same length"):
        be.score([1.0, 2.0], [1.0], be.DistanceMetric.Dot)


def test_distance_metrics(be):                                 # utilsvector.rs:16-35
    v1, v2 = [1.0, 2.0, 3.0], [4.0, 5.0, 6.0]
    cosine = be.score(v1, v2, be.DistanceMetric.Cosine)
    assert 0.0 <= cosine <= 2.0
    assert be.score(v1, v2, be.DistanceMetric.Dot) > 0.0
    assert be.score(v1, v2, be.DistanceMetric.Euclidean) > 0.0


def test_error_display(be):                                    # utilsvector.rs:38-51
    # VectorLengthMismatch Display text (errors.rs:13) as produced through the ABI.
    h = be.HNSWIndex(be.DistanceMetric.Euclidean, 16, 50, 16, 3)
    with pytest.raises(be.VectorLengthMismatch) as e:
        h.insert(1, [1.0, 2.0, 3.0, 4.0, 5.0])
    assert "expected 3" in str(e.value) and "got 5" in str(e.value)


def test_cosine_distance_exact(be):                            # utilsvector.rs:53-60
    assert abs(be.score([1.0, 0.0], [0.0, 1.0], be.DistanceMetric.Cosine) - 1.0) < 1e-6


def test_known_values_bitwise(be):
    """KATs computed by hand from metric.rs:15-35 in fp32."""
    import numpy as np
    f = np.float32
    assert be.score([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], be.DistanceMetric.Dot) == 32.0
    assert f(be.score([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], be.DistanceMetric.Euclidean)) == np.sqrt(f(27.0))
    na, nb = np.sqrt(f(14.0)), np.sqrt(f(77.0))
    expect = f(1.0) - f(32.0) / (na * nb + f(1e-10))
    assert f(be.score([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], be.DistanceMetric.Cosine)) == expect
