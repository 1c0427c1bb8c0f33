"""Const searches on one segment from several host threads (include/vdbmi.h: "const searches on one handle may run
from several host threads"): the compiled-filter cache is shared, bounded and must never hand out bitmaps that
another thread frees."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _segment(vdb, n=3000, d=16):
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    seg = vdb.Segment(vdb.HNSWIndex(2, 8, 32, 16, d, seed=4, filter_mode=vdb.FilterMode.Masked))
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 40)); p.set("half", vdb.PayloadValue.Int(i % 2)); pay.append(p)
    seg.insert_batch(x, pay)
    return seg, x


def _run_threads(workers):
    errors = []
    def wrap(fn):
        def run():
            try: fn()
            except BaseException as e:      # noqa: BLE001 - reported by the main thread
                errors.append(e)
        return run
    ts = [threading.Thread(target=wrap(w)) for w in workers]
    for t in ts: t.start()
    for t in ts: t.join()
    if errors: raise errors[0]


def test_two_threads_with_different_filters_and_a_cache_smaller_than_the_filter_set(vdb):
    seg, x = _segment(vdb)
    q = x[:30] + 0.01
    filters = [vdb.Filter.Match("cat", vdb.PayloadValue.Int(c)) for c in range(40)]        # more than the cache holds
    want = [seg.search_batch(q, 5, filter=f) for f in filters]

    def worker(offset):
        def run():
            for rep in range(3):
                for c in range(40):
                    i = (c + offset) % 40
                    got, gotn = seg.search_batch(q, 5, filter=filters[i])
                    assert np.array_equal(gotn, want[i][1]) and np.array_equal(got["id"], want[i][0]["id"]), i
        return run
    _run_threads([worker(0), worker(20)])


def test_two_threads_compile_the_same_new_filter(vdb):
    seg, x = _segment(vdb)
    q = x[:30] + 0.01
    for c in range(8):
        f = vdb.Filter.And([vdb.Filter.Match("half", vdb.PayloadValue.Int(c % 2)), vdb.Filter.Compare("cat", vdb.ScalarComparisonOp.Lt, vdb.PayloadValue.Int(10 + c))])
        res = [None, None]
        def worker(slot):
            def run(): res[slot] = seg.search_batch(q, 5, filter=f)
            return run
        _run_threads([worker(0), worker(1)])
        assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0]["id"], res[1][0]["id"])
        n_pass, n_err = seg.compile_filter(f)
        assert n_pass == sum(1 for i in range(3000) if i % 2 == c % 2 and i % 40 < 10 + c) and n_err == 0
