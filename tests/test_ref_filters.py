"""Restates /root/reference/tests/filters.rs — host logic, runs for the oracle and the product library."""


def test_match_filter_true(hostbe):                                 # filters.rs:4-16
    b = hostbe
    p = b.Payload(); p.set("kind", b.PayloadValue.Str("robot"))
    assert b.evaluate_filter(b.Filter.Match("kind", b.PayloadValue.Str("robot")), p) is True


def test_match_filter_false(hostbe):                                # filters.rs:18-30
    b = hostbe
    p = b.Payload(); p.set("kind", b.PayloadValue.Str("robot"))
    assert b.evaluate_filter(b.Filter.Match("kind", b.PayloadValue.Str("animal")), p) is False


def test_not_filter(hostbe):                                        # filters.rs:32-44
    b = hostbe
    p = b.Payload(); p.set("status", b.PayloadValue.Str("inactive"))
    assert b.evaluate_filter(b.Filter.Not(b.Filter.Match("status", b.PayloadValue.Str("active"))), p) is True


def test_and_or_filter(hostbe):                                     # filters.rs:46-71
    b = hostbe; PV, F = b.PayloadValue, b.Filter
    p = b.Payload(); p.set("kind", PV.Str("robot")); p.set("active", PV.Bool(True))
    f = F.And([F.Or([F.Match("kind", PV.Str("robot")), F.Match("kind", PV.Str("animal"))]),
               F.Match("active", PV.Bool(True))])
    assert b.evaluate_filter(f, p) is True


def test_short_circuit_and_error_propagation(hostbe):               # filters.rs:44-66 semantics
    import pytest
    b = hostbe; PV, F, Op = b.PayloadValue, b.Filter, b.ScalarComparisonOp
    p = b.Payload(); p.set("a", PV.Int(1))
    missing = F.Compare("nope", Op.Eq, PV.Int(1))
    with pytest.raises(b.InvalidPayload, match="Missing field: nope"):
        b.evaluate_filter(missing, p)
    # And stops at the first false child, so the erroring child is never evaluated
    assert b.evaluate_filter(F.And([F.Match("a", PV.Int(2)), missing]), p) is False
    with pytest.raises(b.InvalidPayload):
        b.evaluate_filter(F.And([F.Match("a", PV.Int(1)), missing]), p)
    assert b.evaluate_filter(F.Or([F.Match("a", PV.Int(1)), missing]), p) is True
    with pytest.raises(b.InvalidPayload):
        b.evaluate_filter(F.Or([missing, F.Match("a", PV.Int(1))]), p)
    with pytest.raises(b.InvalidPayload, match="Type mismatch for field: a"):
        b.evaluate_filter(F.Compare("a", Op.Eq, PV.Float(1.0)), p)      # Int vs Float: no coercion
    assert b.evaluate_filter(F.Match("missing", PV.Int(1)), p) is False  # Match on a missing key is false, not an error
