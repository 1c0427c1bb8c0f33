"""Nodes expanded per step of the layer-0 beam (vdb_options.beam_width) against rows scored, recall and time, on the
benchmark's 1M x 128 index: one graph, imported into one index per width.
Usage: python tools/width_probe.py [ef ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    efs = [int(x) for x in sys.argv[1:]] or [144]
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
    g = index.export_graph()
    d_q = torch.from_numpy(queries).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = [(torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev), torch.zeros(Q, dtype=torch.int32, device=dev),
             torch.zeros(Q * 4, dtype=torch.int32, device=dev)) for _ in range(2)]
    e_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev); e_n = torch.zeros(Q, dtype=torch.int32, device=dev)
    vdb.check(vdb.fn.index_exact_knn_batch_dev(index._h, d_q.data_ptr(), Q, K, e_out.data_ptr(), e_n.data_ptr(), streams[0].cuda_stream))
    torch.cuda.synchronize()
    exact = e_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K)
    del index
    for width in (1, 2, 3, 4):
        p = vdb.HNSWIndex(vdb.DistanceMetric(2), 16, 64, 16, D, seed=SEED, device=0, arith=vdb.Arith.Fast, beam_width=width)
        p.import_graph(g)
        p.sync_device()
        for ef in efs:
            def launch(i):
                o, n, s = bufs[i]
                vdb.check(vdb.fn.index_search_batch_dev(p._h, d_q.data_ptr(), Q, K, ef, o.data_ptr(), n.data_ptr(), s.data_ptr(), streams[i].cuda_stream))
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
            ann = bufs[0][0].cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K); ann_n = bufs[0][1].cpu().numpy().view(np.uint32)
            r = bench.recall_at_k(ann, ann_n, exact, K)
            print(f"width {width} ef {ef}: recall {r:.4f} rows/query {st['rows_scored'].mean():.0f} expanded {st['expanded'].mean():.0f} "
                  f"(p99 {np.percentile(st['expanded'], 99):.0f}, max {st['expanded'].max()})  lone {lone:.3f} ms  two streams {two:.3f} ms/step", flush=True)
        del p


if __name__ == "__main__":
    main()
