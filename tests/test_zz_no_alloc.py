"""vdb_index_exact_knn_batch_dev_noalloc: the exact k-NN entry point that allocates nothing (its candidate pass works
in a scratch array inside the library's code object) against vdb_index_exact_knn_batch_dev, record for record, on two
calls in a row of which the second is smaller; sizes beyond the scratch array and pending edits are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _index(vdb, n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, size=(n, d)).astype(np.float32)
    x[100:140] = x[3]                                          # a tie class that is ranked by id
    ids = (np.arange(n, dtype=np.uint64)[::-1] * 2 + 7).copy()
    dele = np.zeros(n, np.uint8); dele[::11] = 1
    h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, d)
    h.import_graph(dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=x, deleted=dele, layers=[], entry=int(ids[1]), current_max_level=0))
    h.sync_device()
    return h, x


def _both(vdb, torch, h, q, k):
    dev = torch.device("cuda", 0)
    nq = len(q)
    dq = torch.from_numpy(q).to(dev)
    outs = []
    for fn in (vdb.fn.index_exact_knn_batch_dev, vdb.fn.index_exact_knn_batch_dev_noalloc):
        out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        vdb.check(fn(h._h, dq.data_ptr(), nq, k, out.data_ptr(), cnt.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append((out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(nq, k), cnt.cpu().numpy().view(np.uint32), free0 - torch.cuda.mem_get_info()[0]))
    return outs


def test_exact_knn_without_allocation_matches_the_allocating_entry_point(vdb):
    torch = pytest.importorskip("torch")
    big, xb = _index(vdb, 60_000, 64, 1)
    small, xs = _index(vdb, 5_000, 64, 2)
    rng = np.random.default_rng(9)
    q1 = rng.normal(0, 1, size=(300, 64)).astype(np.float32); q1[0] = xb[3]
    q2 = rng.normal(0, 1, size=(40, 64)).astype(np.float32); q2[0] = xs[3]
    for h, q, k in ((big, q1, 10), (small, q2, 10), (big, q1[:17], 20)):     # the second and third calls are smaller than the first
        (ref, refn, _), (got, gotn, grew) = _both(vdb, torch, h, q, k)
        assert np.array_equal(refn, gotn) and (gotn == k).all()
        assert np.array_equal(ref["id"], got["id"])
        assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32))
        assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
        assert grew <= 0, f"device memory in use grew by {grew} bytes across a call that must not allocate"


def test_sizes_beyond_the_scratch_array_and_pending_edits_are_refused(vdb):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    h, x = _index(vdb, 5_000, 8, 3)
    k = 10
    nq = (4 << 20) // 16                                       # nq * 16 candidate words alone fill the array: with 2 * rows on top it cannot fit
    dq = torch.zeros(nq * 8, dtype=torch.float32, device=dev)
    out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    with pytest.raises(vdb.ArgumentError, match="static scratch"):
        vdb.check(vdb.fn.index_exact_knn_batch_dev_noalloc(h._h, dq.data_ptr(), nq, k, out.data_ptr(), cnt.data_ptr(), None))
    h.mark_deleted(7 + 2 * 10)
    with pytest.raises(vdb.ArgumentError, match="sync_device"):
        vdb.check(vdb.fn.index_exact_knn_batch_dev_noalloc(h._h, dq.data_ptr(), 16, k, out.data_ptr(), cnt.data_ptr(), None))
    h.sync_device()
    vdb.check(vdb.fn.index_exact_knn_batch_dev_noalloc(h._h, dq.data_ptr(), 16, k, out.data_ptr(), cnt.data_ptr(), None))
    with pytest.raises(vdb.ArgumentError):
        vdb.check(vdb.fn.index_exact_knn_batch_dev_noalloc(h._h, dq.data_ptr(), 16, 40, out.data_ptr(), cnt.data_ptr(), None))
    torch.cuda.synchronize()
