"""Restates /root/reference/tests/payload.rs (12 tests) — host logic, oracle and product library."""
import pytest


def test_scalar_comparisons(hostbe):                                # payload.rs(tests):5-27
    b = hostbe; PV, Op = b.PayloadValue, b.ScalarComparisonOp
    assert PV.Int(10).compare_scalar(Op.Lt, PV.Int(20)) is True
    assert PV.Float(1.0).compare_scalar(Op.Eq, PV.Float(1.0)) is True
    assert PV.Str("abc").compare_scalar(Op.Lt, PV.Str("xyz")) is True
    assert PV.Bool(True).compare_scalar(Op.Neq, PV.Bool(False)) is True
    assert PV.Bool(True).compare_scalar(Op.Eq, PV.Str("true")) is None
    assert PV.Bool(True).compare_scalar(Op.Lt, PV.Bool(False)) is None      # payload.rs:179-183
    assert PV.Int(1).compare_scalar(Op.Eq, PV.Float(1.0)) is None
    nan = float("nan")
    assert PV.Float(nan).compare_scalar(Op.Eq, PV.Float(nan)) is True       # OrderedFloat total order
    assert PV.Float(nan).compare_scalar(Op.Gt, PV.Float(1e300)) is True
    assert PV.Float(-0.0).compare_scalar(Op.Eq, PV.Float(0.0)) is True


def test_list_contains(hostbe):                                     # :29-52
    b = hostbe; PV, K = b.PayloadValue, b.ListQueryKind
    assert PV.ListInt([1, 2, 3]).evaluate_list_query(K.Contains, value=PV.Int(2)) is True
    assert PV.ListFloat([0.1, 0.2]).evaluate_list_query(K.Contains, value=PV.Float(0.3)) is False
    assert PV.ListStr(["a", "b"]).evaluate_list_query(K.Contains, value=PV.Str("a")) is True
    assert PV.ListBool([True, False]).evaluate_list_query(K.Contains, value=PV.Bool(False)) is True


def test_list_element_compare(hostbe):                              # :54-77
    b = hostbe; PV, K, Op = b.PayloadValue, b.ListQueryKind, b.ScalarComparisonOp
    assert PV.ListInt([5, 10, 15]).evaluate_list_query(K.ElementCompare, n=2, op=Op.Gt, value=PV.Int(10)) is True
    assert PV.ListFloat([0.1, 0.5]).evaluate_list_query(K.ElementCompare, n=1, op=Op.Lte, value=PV.Float(0.5)) is True
    assert PV.ListStr(["a", "b"]).evaluate_list_query(K.ElementCompare, n=0, op=Op.Eq, value=PV.Str("a")) is True
    assert PV.ListBool([True, False]).evaluate_list_query(K.ElementCompare, n=1, op=Op.Eq, value=PV.Bool(False)) is True


def test_list_length(hostbe):                                       # :79-91
    b = hostbe; PV, K, Op = b.PayloadValue, b.ListQueryKind, b.ScalarComparisonOp
    v = PV.ListStr(["a", "b", "c"])
    assert v.evaluate_list_query(K.Length, op=Op.Eq, n=3) is True
    assert v.evaluate_list_query(K.Length, op=Op.Gt, n=1) is True
    assert PV.Int(3).evaluate_list_query(K.Length, op=Op.Eq, n=3) is None


def test_list_equality(hostbe):                                     # :93-108
    b = hostbe; PV, K = b.PayloadValue, b.ListQueryKind
    a = PV.ListInt([1, 2, 3])
    assert a.evaluate_list_query(K.Equals, value=PV.ListInt([1, 2, 3])) is True
    assert a.evaluate_list_query(K.Equals, value=PV.ListInt([1, 2])) is False
    assert a.evaluate_list_query(K.Equals, value=PV.ListFloat([1.0, 2.0, 3.0])) is None


def test_payload_set_and_get(hostbe):                               # :110-115
    b = hostbe
    p = b.Payload(); p.set("flag", b.PayloadValue.Bool(True))
    assert p.get("flag") == b.PayloadValue.Bool(True)


def test_payload_compare_field(hostbe):                             # :117-130
    b = hostbe; PV, Op = b.PayloadValue, b.ScalarComparisonOp
    p = b.Payload(); p.set("x", PV.Int(42))
    assert p.compare_field("x", Op.Gte, PV.Int(40)) is True
    with pytest.raises(b.InvalidPayload, match="Missing field: y"):
        p.compare_field("y", Op.Eq, PV.Int(1))
    with pytest.raises(b.InvalidPayload, match="Type mismatch for field: x"):
        p.compare_field("x", Op.Eq, PV.Str("forty-two"))


def test_payload_compare_field_liststr(hostbe):                     # payload.rs:50-90
    b = hostbe; PV, Op = b.PayloadValue, b.ScalarComparisonOp
    p = b.Payload(); p.set("tags", PV.ListStr(["a", "b"]))
    assert p.compare_field("tags", Op.Eq, PV.Str("a")) is True
    assert p.compare_field("tags", Op.Neq, PV.Str("a")) is False
    assert p.compare_field("tags", Op.Neq, PV.Str("z")) is True
    assert p.compare_field("tags", Op.Eq, PV.ListStr(["a", "b"])) is True
    assert p.compare_field("tags", Op.Neq, PV.ListStr(["a"])) is True
    with pytest.raises(b.InvalidPayload, match="Invalid operation for ListStr and Str"):
        p.compare_field("tags", Op.Lt, PV.Str("a"))
    with pytest.raises(b.InvalidPayload, match="Invalid operation for ListStr"):
        p.compare_field("tags", Op.Gt, PV.ListStr(["a"]))
    with pytest.raises(b.InvalidPayload, match="Type mismatch for field: tags"):
        p.compare_field("tags", Op.Eq, PV.Int(1))


def test_payload_evaluate_list_field_errors(hostbe):                # :132-154
    b = hostbe; PV, K = b.PayloadValue, b.ListQueryKind
    p = b.Payload(); p.set("tags", PV.ListStr(["a", "b"]))
    assert p.evaluate_list_field("tags", K.Contains, value=PV.Str("a")) is True
    with pytest.raises(b.InvalidPayload, match="Missing field: nonexistent"):
        p.evaluate_list_field("nonexistent", K.Contains, value=PV.Str("a"))
    with pytest.raises(b.InvalidPayload, match="Invalid list operation on field: tags"):
        p.evaluate_list_field("tags", K.Contains, value=PV.Float(1.0))


def test_payload_overwrite(hostbe):                                 # :156-163
    b = hostbe
    p = b.Payload(); p.set("val", b.PayloadValue.Int(10)); p.set("val", b.PayloadValue.Int(20))
    assert p.get("val") == b.PayloadValue.Int(20)


def test_payload_default_behavior(hostbe):                          # :165-169
    assert hostbe.Payload().get("missing") is None


def test_list_element_compare_out_of_bounds(hostbe):                # :171-177
    b = hostbe; PV, K, Op = b.PayloadValue, b.ListQueryKind, b.ScalarComparisonOp
    assert PV.ListInt([1, 2]).evaluate_list_query(K.ElementCompare, n=5, op=Op.Eq, value=PV.Int(1)) is None


def test_list_element_compare_wrong_type(hostbe):                   # :179-185
    b = hostbe; PV, K, Op = b.PayloadValue, b.ListQueryKind, b.ScalarComparisonOp
    assert PV.ListStr(["a"]).evaluate_list_query(K.ElementCompare, n=0, op=Op.Eq, value=PV.Bool(True)) is None
    assert PV.ListBool([True]).evaluate_list_query(K.ElementCompare, n=0, op=Op.Lt, value=PV.Bool(True)) is None
