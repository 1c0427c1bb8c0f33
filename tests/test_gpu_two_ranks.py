"""Two ranks, two shards, ONE GPU: the sharded search path with the HIP backend and a real process group.

RCCL refuses two ranks on one device, so the ranks exchange through `gloo` (ShardedIndex stages the lists through
host memory for a host-side backend); everything else is the multi-GPU path as it runs under torchrun: contiguous slot
ranges (shard_range), one sub-graph per rank built and searched by libvdbmi on the GPU, one all-gather of the per-shard
top-k, the k-way merge kernel (vdb_topk_merge_dev) on every rank.  The reference has one in-memory segment
(src/segment/segment.rs:11-18) and no counterpart; expected values are the exact k-NN of ONE index over the whole set
(ids and keys bit for bit) and the (sort_key, id) merge of the shards' own graph-search lists.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N, D, Q, K, EF = 20_000, 48, 96, 10, 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dataset():
    rng = np.random.default_rng(4242)
    base = rng.normal(size=(N, D)).astype(np.float32)
    base[N // 2 + 77] = base[5]                             # exact tie across the two shards
    queries = rng.normal(size=(Q, D)).astype(np.float32)
    queries[0] = base[5]
    return base, queries


def _rank_main(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import vectordb_amd
    from vectordb_amd.sharded import ShardedIndex, shard_range
    vdb = vectordb_amd.api()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base, queries = _dataset()
        lo, hi = shard_range(N, world, rank)
        idx = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=11 + rank, device=0, arith=vdb.Arith.Fast)
        idx.insert_batch(np.arange(lo, hi, dtype=np.uint64), base[lo:hi])
        sh = ShardedIndex(vdb, idx, device=dev)
        assert (sh.rank, sh.world) == (rank, world)
        # host-buffer path
        ex, exn = sh.exact_knn_batch(queries, K)
        loc, locn = idx.search_batch(queries, K, ef_search=EF)
        mg, mgn = sh.search_batch(queries, K, ef_search=EF)
        # device-buffer path: search on a stream of its own, exchange + merge ordered behind it by an event
        st = torch.cuda.Stream(device=dev)
        d_q = torch.from_numpy(queries).to(dev)
        d_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev); d_n = torch.zeros(Q, dtype=torch.int32, device=dev)
        d_st = torch.zeros(Q * 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        vdb.check(vdb.fn.index_search_batch_dev(idx._h, d_q.data_ptr(), Q, K, EF, d_out.data_ptr(), d_n.data_ptr(), d_st.data_ptr(), st.cuda_stream))
        ev = torch.cuda.Event(); ev.record(st)
        torch.cuda.current_stream(dev).wait_event(ev)
        m_out, m_n = sh.merge_dev(d_out, d_n, Q, K)
        torch.cuda.synchronize()
        dmg = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K); dmgn = m_n.cpu().numpy().view(np.uint32)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), ex=ex, exn=exn, loc=loc, locn=locn, mg=mg, mgn=mgn, dmg=dmg, dmgn=dmgn)
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_and_merge_their_shards(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]

    # every rank holds the same merged answer, host path and device path alike
    for key in ("ex", "exn", "mg", "mgn", "dmg", "dmgn"):
        assert r[0][key].tobytes() == r[1][key].tobytes(), key
    assert np.array_equal(r[0]["mgn"], r[0]["dmgn"])
    for q in range(Q):
        n = int(r[0]["mgn"][q])
        assert r[0]["mg"][q][:n].tobytes() == r[0]["dmg"][q][:n].tobytes()

    # exact k-NN over the shards == exact k-NN of one index over the whole set, bit for bit
    import vectordb_amd
    vdb = vectordb_amd.api()
    base, queries = _dataset()
    whole = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, D, seed=3, device=0, arith=vdb.Arith.Fast)
    whole.insert_batch(np.arange(N, dtype=np.uint64), base)
    ref, refn = whole.exact_knn_batch(queries, K)
    assert np.array_equal(refn, r[0]["exn"])
    assert np.array_equal(ref["id"], r[0]["ex"]["id"])
    assert ref["sort_key"].tobytes() == r[0]["ex"]["sort_key"].tobytes()
    assert list(r[0]["ex"]["id"][0][:2]) == [5, N // 2 + 77]      # equal keys from different shards: smaller id first

    # merged graph search == (sort_key, id) merge of the two shards' own lists
    for q in range(Q):
        rows = np.concatenate([r[k]["loc"][q][: r[k]["locn"][q]] for k in range(world)])
        order = np.lexsort((rows["id"], rows["sort_key"]))[:K]
        n = int(r[0]["mgn"][q])
        assert n == len(order)
        assert np.array_equal(rows["id"][order], r[0]["mg"]["id"][q][:n])
        assert rows["raw_score"][order].tobytes() == r[0]["mg"]["raw_score"][q][:n].tobytes()
    # and it finds the neighbours: recall@10 of the merged graph search against the exact lists
    hits = sum(len(np.intersect1d(r[0]["mg"]["id"][q][:K], ref["id"][q][:K])) for q in range(Q))
    assert hits / (Q * K) >= 0.95
