"""Restates /root/reference/tests/inverted_index.rs (7 tests) — host logic, oracle and product library."""


def test_index_insert_and_query(hostbe):                            # inverted_index.rs:7-36
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex()
    p = b.Payload()
    p.set("category", PV.Str("fruit")); p.set("rank", PV.Int(1)); p.set("confidence", PV.Float(0.95)); p.set("active", PV.Bool(True))
    idx.insert(42, p); idx.insert(43, p)
    assert idx.query_exact("category", PV.Str("fruit")) == {42, 43}
    assert idx.query_exact("rank", PV.Int(1)) == {42, 43}
    assert idx.query_exact("confidence", PV.Float(0.95)) == {42, 43}
    assert idx.query_exact("active", PV.Bool(True)) == {42, 43}


def test_index_removal(hostbe):                                     # :38-58
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex(); p = b.Payload(); p.set("rank", PV.Int(99))
    idx.insert(1, p); idx.insert(2, p)
    assert len(idx.query_exact("rank", PV.Int(99))) == 2
    idx.remove(1, p)
    assert idx.query_exact("rank", PV.Int(99)) == {2}
    idx.remove(2, p)
    assert idx.query_exact("rank", PV.Int(99)) is None
    assert idx.all_for_key("rank") is None                           # stores.rs:47-49 prunes the key


def test_non_indexed_types(hostbe):                                 # :60-75
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex(); p = b.Payload()
    p.set("list", PV.ListStr(["a", "b"])); p.set("numbers", PV.ListInt([1, 2, 3]))
    idx.insert(99, p)
    assert idx.query_exact("list", PV.Str("a")) is None
    assert idx.query_exact("numbers", PV.Int(1)) is None
    assert idx.all_for_key("list") is None and idx.all_for_key("numbers") is None
    assert idx.query_exact("list", PV.ListStr(["a", "b"])) is None   # stores.rs:56-58


def test_all_for_key(hostbe):                                       # :77-96
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex()
    for pid, c in ((1, "red"), (2, "blue"), (3, "red")):
        p = b.Payload(); p.set("color", PV.Str(c)); idx.insert(pid, p)
    assert idx.all_for_key("color") == {1, 2, 3}


def test_duplicate_inserts_are_idempotent(hostbe):                  # :98-110
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex(); p = b.Payload(); p.set("kind", PV.Str("apple"))
    idx.insert(1, p); idx.insert(1, p)
    assert idx.query_exact("kind", PV.Str("apple")) == {1}


def test_insert_same_key_different_values(hostbe):                  # :112-133
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex()
    p1 = b.Payload(); p1.set("group", PV.Str("A")); p2 = b.Payload(); p2.set("group", PV.Str("B"))
    idx.insert(1, p1); idx.insert(2, p2)
    assert idx.query_exact("group", PV.Str("A")) == {1}
    assert idx.query_exact("group", PV.Str("B")) == {2}


def test_query_nonexistent_key_or_value(hostbe):                    # :135-145
    b = hostbe; PV = b.PayloadValue
    idx = b.PayloadIndex(); p = b.Payload(); p.set("status", PV.Str("ok")); idx.insert(1, p)
    assert idx.query_exact("nonexistent", PV.Str("nope")) is None
    assert idx.query_exact("status", PV.Str("error")) is None
