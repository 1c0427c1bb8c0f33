"""How much of the search kernel's HBM traffic is re-reads that a locality-aware order of the batch would keep in cache?

Builds the benchmark's 1M x 128 index, then times the same 10k-query batch in several orders (results are per query, so the
order changes nothing but which walks are resident together):
  original      as generated (centres drawn at random per query)
  by_centre     sorted by the nearest generator centre (perfect knowledge: an upper bound for any on-line key)
  striped       the same, dealt out so that block b of the launch (which runs on XCD b mod 8) takes position
                (b mod 8) * Q/8 + b div 8 of the sorted order: each XCD's L2 sees one contiguous run of it
Usage: python tools/query_order_probe.py [ef]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ef = int(sys.argv[1]) if len(sys.argv) > 1 else 144
    import torch
    import vectordb_amd
    vdb = vectordb_amd.api()
    cfg = bench.CONFIGS["c2"]
    N, D, Q, K, SEED = cfg["n"], cfg["dim"], cfg["queries"], 10, cfg["seed"]
    dev = torch.device("cuda", 0)
    centres = np.random.default_rng(SEED).uniform(0, 128, size=(cfg["centres"], D)).astype(np.float32)
    base = bench.clustered(N, D, SEED + 1, centres)
    queries = bench.clustered(Q, D, SEED + 1000, centres)
    ids = np.arange(N, dtype=np.uint64)
    t0 = time.time()
    index = vdb.HNSWIndex(vdb.DistanceMetric(2), 16, 64, 16, D, seed=SEED, device=0, arith=vdb.Arith.Fast)
    for s in range(0, N, 250_000):
        index.insert_batch(ids[s:s + 250_000], base[s:s + 250_000])
    print(f"built in {time.time()-t0:.1f}s", flush=True)

    d2 = (queries ** 2).sum(1)[:, None] - 2.0 * queries @ centres.T + (centres ** 2).sum(1)[None, :]
    cid = d2.argmin(1)
    by_centre = np.argsort(cid, kind="stable")
    b = np.arange(Q)
    per = (Q + 7) // 8
    pos = (b % 8) * per + b // 8
    striped = by_centre[np.minimum(pos, Q - 1)] if Q % 8 else by_centre[pos]
    orders = {"original": np.arange(Q), "by_centre": by_centre, "striped": striped}
    rng = np.random.default_rng(1)
    orders["shuffled"] = rng.permutation(Q)

    def stripe(o):
        return o[pos] if Q % 8 == 0 else o[np.minimum(pos, Q - 1)]

    # keys a search could compute for itself: where the greedy descent (hnsw.rs:538-541) stands after each upper layer
    t0 = time.time()
    top = index.current_max_level()
    ep = index.get_entry_point()
    at = np.zeros((Q, top + 1), dtype=np.uint64)
    for qi in range(Q):
        cur = ep
        for l in range(top, 0, -1):
            cur = index.greedy_search_layer_unfiltered(queries[qi], cur, l)
            at[qi, l] = cur
    print(f"descents in {time.time()-t0:.1f}s; max level {top}; distinct nodes per level: "
          + ", ".join(f"L{l}: {len(np.unique(at[:, l]))}" for l in range(top, 0, -1)), flush=True)
    for l in range(1, min(top, 3) + 1):
        # (node reached at level l, then the finer levels below it)
        keys = [at[:, j] for j in range(1, l + 1)]          # lexsort: last key is the primary one
        orders[f"descent_L{l}_striped"] = stripe(np.lexsort(keys))
    # same-centre queries that agree on the key (how well the key finds cluster mates)
    for l in range(1, min(top, 3) + 1):
        same = 0
        o = np.lexsort([at[:, j] for j in range(1, l + 1)])
        same = float((cid[o][1:] == cid[o][:-1]).mean())
        print(f"  L{l} order: neighbours in the order share a centre {same:.2f} of the time (by_centre: {(cid[by_centre][1:] == cid[by_centre][:-1]).mean():.2f})", flush=True)

    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = [(torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev), torch.zeros(Q, dtype=torch.int32, device=dev),
             torch.zeros(Q * 4, dtype=torch.int32, device=dev)) for _ in range(2)]
    for name, order in orders.items():
        d_q = torch.from_numpy(np.ascontiguousarray(queries[order])).to(dev)
        torch.cuda.synchronize()

        def launch(i):
            o, n, s = bufs[i]
            vdb.check(vdb.fn.index_search_batch_dev(index._h, d_q.data_ptr(), Q, K, ef, o.data_ptr(), n.data_ptr(), s.data_ptr(), streams[i].cuda_stream))

        for _ in range(3):
            launch(0)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        for _ in range(20):
            launch(0)
        e1.record(streams[0])
        torch.cuda.synchronize()
        lone = e0.elapsed_time(e1) / 20
        for i in range(4):
            launch(i % 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            launch(i % 2)
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) * 1e3 / 40
        st = bufs[0][2].cpu().numpy().view(vdb.STATS_DTYPE).reshape(Q)
        print(f"{name:10s} lone {lone:.3f} ms  two streams {two:.3f} ms/step ({Q/two/1e3:.2f} M q/s)  rows/query {st['rows_scored'].mean():.0f}", flush=True)


if __name__ == "__main__":
    main()
