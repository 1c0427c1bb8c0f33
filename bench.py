#!/usr/bin/env python3
"""bench.py — queries/sec at recall@10 >= 0.95 on the SIFT1M-shape synthetic set (1M x 128 fp32, L2,
10k-query batch): the segment search path measured through the C ABI.

One "step" = one pass of the batched HNSW search (greedy descent + layer-0 ef-beam, one wavefront per
query) over the whole 10k-query batch, queries and results resident in HBM.  The graph is built on
the GPU by the library's batched inserter, the ground truth by its exact k-NN kernel; the ef sweep
picks the smallest beam that reaches recall@10 >= 0.95 and `value` is measured at that ef.

  python bench.py --gpus N --steps K --warmup W
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns its own 1M-vector shard
and sub-graph, searches the same query batch, and the per-shard top-k lists are merged after one
RCCL all-gather; a unit is one query searched against one shard, so value = N * Q * K / time.

Other configurations of the same path (the default line stays on the one above):
  --config c3   GIST-shape: 1M x 960 fp32 cosine, 1000-query batch
  --config c4   10M x 768 dot product with payload filters (1 %, 5 %, 10 %, 50 % of the rows pass), 10k-query
                device-buffer batches through vdb_segment_search_batch_dev; one JSON line, `value` at the 10 % filter
  --config c5   Deep100M-shape: 96-d L2, one 12.5M-row shard per rank (weak: 25M / 50M / 100M rows at 2 / 4 / 8 GPUs);
                --c5-total 100000000 fixes the total instead (strong: every rank holds total / N rows)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, datasheet
KERNEL_SOURCES = ("search_kernel.hpp", "search_team.hpp", "device_common.hpp", "kernels.hpp", "common.hpp", "search_plain.hip")


def kernel_sources_hash():
    """Identifies the code of the timed kernel: counter traffic measured on other code is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "vectordb_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def clustered(n, d, seed, centres, sigma=12.0, chunk=131072):
    """SIFT-like synthetic rows: Gaussian blobs around `centres`, clipped at 0."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        c = centres[rng.integers(0, len(centres), size=e - s)]
        out[s:e] = np.clip(c + rng.standard_normal((e - s, d), dtype=np.float32) * sigma, 0, None)
    return out


def recall_at_k(ann, ann_n, exact, k):
    hits = 0
    for a, n, e in zip(ann, ann_n, exact):
        ek = e["sort_key"][k - 1]
        got = a[: min(n, k)]
        # ids in the exact top-k, or tied with its k-th key within 1e-6 relative
        ok = np.isin(got["id"], e["id"][:k]) | (np.abs(got["sort_key"] - ek) <= 1e-6 * max(abs(ek), 1e-30))
        hits += int(ok.sum())
    return hits / (len(ann) * k)


# name -> rows per rank, dimension, queries, metric (DistanceMetric value), Gaussian centres, generator seed, ef candidates
CONFIGS = {
    "c1": dict(n=10_000, dim=128, queries=1_000, metric=0, centres=0, seed=0xC1, efs=[64], streams=2,
               metric_name="queries/sec, 10k x 128 cosine, HNSW M=16 ef=64 (the reference's own test shape), recall@10 reported", shape="tests/hnsw.rs-shape"),
    "c2u": dict(n=1_000_000, dim=128, queries=10_000, metric=2, centres=0, uniform=(0.0, 1.0), seed=0xC2, efs=[64, 128, 256, 512],
                metric_name="queries/sec at recall@10>=0.95 (1M x 128 L2, uniform(0, 1) stress variant)", shape="uniform stress"),
    "c2": dict(n=1_000_000, dim=128, queries=10_000, metric=2, centres=4096, seed=0xC2, efs=[16, 32, 64, 96, 128, 144, 160, 176, 192, 224, 256],
               metric_name="queries/sec at recall@10>=0.95 (1M x 128 L2)", shape="SIFT1M-shape"),
    "c3": dict(n=1_000_000, dim=960, queries=1_000, metric=0, centres=1024, seed=0xC3, efs=[64, 128, 192, 256, 384, 416, 448, 480, 512], streams=6,
               metric_name="queries/sec at recall@10>=0.95 (1M x 960 cosine)", shape="GIST-shape"),
    "c5": dict(n=12_500_000, dim=96, queries=10_000, metric=2, centres=4096, seed=0xC5, efs=[32, 64, 128, 256],
               metric_name="queries/sec (100M x 96 L2 over 8 shards), recall@10 reported", shape="Deep100M-shape"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c2u", "c3", "c4", "c5"])
    ap.add_argument("--c5-total", type=int, default=0, help="c5: total rows over all ranks (strong scaling); 0 = 12.5M per rank")
    ap.add_argument("--n", type=int, default=0, help="rows per rank (0 = the configuration's)")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=0, help="0 = sweep and pick the smallest ef with recall >= target")
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct query batches the timed steps rotate through (step i searches batch i mod B), so that no step "
                         "finds the rows of its own previous run in a cache; 1 = replay one batch")
    ap.add_argument("--cpu-queries", type=int, default=0, help="queries of the CPU baseline sample (0 = all for c2, 1000 otherwise)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the timed steps rotate over (consecutive batches overlap, which hides the tail of "
                         "long-running queries at the end of each launch); 1 = strictly back to back; 0 = the configuration's "
                         "(2; 6 for c3, whose 1000-query launches leave most of the machine idle while their longest walks end)")
    args = ap.parse_args()
    if args.config == "c4":
        return bench_c4(args)
    cfg = CONFIGS[args.config]

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch                                   # before libvdbmi: one HIP runtime per process
    # Rehearsal of the multi-rank step on a box with ONE GPU: VDB_BENCH_REHEARSAL=gloo puts every rank on device 0 and
    # exchanges through gloo (host-staged) instead of RCCL, which refuses two ranks on one device.  Never set by the driver.
    rehearsal = os.environ.get("VDB_BENCH_REHEARSAL", "")
    gpu_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    # VDB_BENCH_EXCHANGE=1 under torchrun with one rank rehearses the multi-GPU step (all-gather + merge) on one GPU
    exchange = world > 1 or (os.environ.get("VDB_BENCH_EXCHANGE") == "1" and "RANK" in os.environ)
    if exchange:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(rehearsal)
        else:
            dist.init_process_group("nccl", device_id=dev)
    import vectordb_amd
    vdb = vectordb_amd.api()

    N, D, Q, K = args.n or cfg["n"], args.dim or cfg["dim"], args.queries or cfg["queries"], args.k
    if args.config == "c5" and args.c5_total:
        N = args.c5_total // world                               # strong scaling: the total is fixed, contiguous id ranges
    METRIC, SEED = cfg["metric"], cfg["seed"]
    M, EF_BUILD = 16, 64
    t0 = time.time()
    if cfg["centres"]:
        centres = np.random.default_rng(SEED).uniform(0, 128, size=(cfg["centres"], D)).astype(np.float32)
        base = clustered(N, D, SEED + 1 + rank, centres)
        batches = [clustered(Q, D, SEED + 1000 + b, centres) for b in range(max(1, args.batches))]
    else:                                                        # c1: uniform(-1, 1) per coordinate, like generate_points of tests/hnsw.rs:12-26
        lo, hi = cfg.get("uniform", (-1.0, 1.0))
        base = np.random.default_rng(SEED + rank).uniform(lo, hi, size=(N, D)).astype(np.float32)
        batches = [np.random.default_rng(SEED + 1000 + b).uniform(lo, hi, size=(Q, D)).astype(np.float32) for b in range(max(1, args.batches))]
    ids = np.arange(N, dtype=np.uint64) + np.uint64(rank) * np.uint64(N)
    queries, NB = batches[0], len(batches)
    log(f"data: {N}x{D} base (shard {rank}/{world}), {NB} batches of {Q} queries in {time.time()-t0:.1f}s")

    # ---- build the shard's graph on the GPU (hnsw.rs:119-208 linking rules, batched rounds) ----
    t0 = time.time()
    index = vdb.HNSWIndex(vdb.DistanceMetric(METRIC), M, EF_BUILD, 16, D, seed=SEED + rank, device=gpu_index, arith=vdb.Arith.Fast)
    step = 250_000
    for s in range(0, N, step):
        index.insert_batch(ids[s:s + step], base[s:s + step])
        log(f"  inserted {min(N, s + step)} / {N} ({time.time()-t0:.1f}s)")
    info = index.device_info()
    build_s = time.time() - t0
    log(f"graph: {info['l0_edges']} layer-0 edges (mean degree {info['l0_edges']/N:.1f}), {info['upper_edges']} upper edges, "
        f"max level {index.current_max_level()}, {info['bytes_total']/2**20:.0f} MiB on device, built in {build_s:.1f}s")

    # ---- device-resident inputs/outputs ----
    stream = torch.cuda.Stream(device=dev)
    d_qs = [torch.from_numpy(b).to(dev) for b in batches]
    d_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_stats = torch.zeros(Q * 4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    sp = stream.cuda_stream

    def search(ef, b=0):
        vdb.check(vdb.fn.index_search_batch_dev(index._h, d_qs[b].data_ptr(), Q, K, ef, d_out.data_ptr(), d_n.data_ptr(), d_stats.data_ptr(), sp))

    def fetch():
        stream.synchronize()
        out = d_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K)
        return out, d_n.cpu().numpy().view(np.uint32), d_stats.cpu().numpy().view(vdb.STATS_DTYPE).reshape(Q)

    from vectordb_amd.sharded import ShardedIndex
    shards = ShardedIndex(vdb, index, device=dev, always_exchange=exchange)

    def merged(out_t, n_t):
        """all-gather per-shard lists and merge them on the GPU (one exchange step)."""
        return shards.merge_dev(out_t, n_t, Q, K)

    # ---- ground truth: exact k-NN on the GPU (strict arithmetic, (key, id) order) ----
    exacts, exact_s = [], 0.0
    e_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev); e_n = torch.zeros(Q, dtype=torch.int32, device=dev)
    for b in range(NB):
        t0 = time.time()
        vdb.check(vdb.fn.index_exact_knn_batch_dev(index._h, d_qs[b].data_ptr(), Q, K, e_out.data_ptr(), e_n.data_ptr(), sp))
        stream.synchronize()
        if b or NB == 1: exact_s += time.time() - t0         # with several batches the first call, which also allocates, is left out
        torch.cuda.synchronize()
        ge_out, ge_n = merged(e_out, e_n)
        torch.cuda.synchronize()
        exacts.append(ge_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K).copy())
    exact_s /= max(1, NB - 1)
    exact = exacts[0]
    log(f"exact k-NN ground truth: {exact_s:.3f}s per batch ({2.0*Q*N*D/exact_s/1e12:.2f} TFLOP/s equivalent)")

    # ---- ef sweep ----
    sweep = {}
    efs = [args.ef] if args.ef else cfg["efs"]
    chosen = None
    for ef in efs:
        per_batch, rows, rows_max, overflow = [], [], 0, 0
        for b in range(NB):                                   # the recall of every batch the timed steps will search
            search(ef, b); stream.synchronize()
            torch.cuda.synchronize()
            m_out, m_n = merged(d_out, d_n)
            torch.cuda.synchronize()
            ann = m_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K); ann_n = m_n.cpu().numpy().view(np.uint32)
            st = d_stats.cpu().numpy().view(vdb.STATS_DTYPE).reshape(Q)
            per_batch.append(recall_at_k(ann, ann_n, exacts[b], K)); rows.append(float(st["rows_scored"].mean()))
            rows_max = max(rows_max, int(st["rows_scored"].max())); overflow += int((st["flags"] & 1).sum())
        r = float(np.mean(per_batch))
        sweep[ef] = dict(recall=round(r, 4), recall_per_batch=[round(x, 4) for x in per_batch], rows_scored=float(np.mean(rows)), overflow=overflow)
        log(f"  ef={ef}: recall@{K}={r:.4f} (batches {min(per_batch):.4f} .. {max(per_batch):.4f}) rows_scored/query={np.mean(rows):.0f} max={rows_max} overflow={overflow}")
        if chosen is None and r >= args.target_recall:
            chosen = ef
            if not args.ef:
                break
    if chosen is None:
        chosen = efs[-1]
        log(f"WARNING: target recall {args.target_recall} not reached; reporting at ef={chosen}")
    ef = chosen

    # ---- timed regions ----
    def barrier():
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    n_streams = max(1, args.streams or cfg.get("streams", 2))
    lanes = [(stream, d_out, d_n, d_stats)]
    for _ in range(n_streams - 1):
        lanes.append((torch.cuda.Stream(device=dev), torch.zeros_like(d_out), torch.zeros_like(d_n), torch.zeros_like(d_stats)))
    xstream = torch.cuda.Stream(device=dev) if exchange else None
    searched = [torch.cuda.Event() for _ in lanes]      # search of the lane's current step is enqueued
    merged_ev = [None for _ in lanes]                   # the exchange that last read the lane's buffers

    def launch(lane, b):
        st_, o_, n_, s_ = lane
        vdb.check(vdb.fn.index_search_batch_dev(index._h, d_qs[b].data_ptr(), Q, K, ef, o_.data_ptr(), n_.data_ptr(), s_.data_ptr(), st_.cuda_stream))

    def one_step(i, replay=False):
        li = i % n_streams
        lane = lanes[li]
        if exchange and merged_ev[li] is not None:
            lane[0].wait_event(merged_ev[li])           # the previous exchange has consumed these buffers
        launch(lane, 0 if replay else i % NB)
        if exchange:
            # exchange + merge on a stream of their own, ordered after this lane's search by an event (no host
            # synchronisation): the next step's search overlaps it on the other lane.  Not the null stream: whatever is
            # recorded there also waits for the other lane's running search, and the lanes fall into lockstep.
            searched[li].record(lane[0])
            xstream.wait_event(searched[li])
            with torch.cuda.stream(xstream):
                merged(lane[1], lane[2])
            merged_ev[li] = torch.cuda.Event(); merged_ev[li].record(xstream)

    # (A) back to back on ONE stream, HIP events on that stream: the kernel's own average duration
    def lone(replay):
        for i in range(args.warmup):
            launch(lanes[0], 0 if replay else i % NB)
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for i in range(args.steps):
            launch(lanes[0], 0 if replay else (args.warmup + i) % NB)
        ev1.record(stream)
        barrier()
        return ev0.elapsed_time(ev1) / args.steps
    kernel_ms = lone(False)
    kernel_ms_replayed = lone(True) if NB > 1 else kernel_ms

    # (B) the timed K steps: whole-job throughput, steps rotating over the streams and over the query batches
    def timed(replay):
        for i in range(args.warmup):
            one_step(i, replay)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i, replay)
        barrier()
        wall = time.perf_counter() - t0
        if exchange:
            tt = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall = float(tt.item())
        return wall
    wall = timed(False)
    wall_replayed = timed(True) if NB > 1 else wall
    ms_per_step = wall * 1e3 / args.steps
    value = world * Q * args.steps / wall

    barrier()
    search(ef, 0)
    out, out_n, st = fetch()
    result = {
        "metric": cfg["metric_name"], "value": value, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if (args.config == "c5" and args.c5_total) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg['shape']} synthetic: {N} x {D} fp32 {('cosine', 'dot', 'L2')[METRIC]} per GPU ({'clustered, ' + str(cfg['centres']) + ' Gaussian centres' if cfg['centres'] else 'uniform coordinates in ' + str(cfg.get('uniform', (-1.0, 1.0)))}), "
                               f"{Q}-query batch, HNSW M={M} ef_construction={EF_BUILD}, top-{K}",
                   "rccl_ranks": shards.rccl_ranks(), "total_rows": N * world,
                   "ef_search": ef, "recall_at_10": sweep[ef]["recall"], "recall_at_10_per_batch": sweep[ef]["recall_per_batch"],
                   "query_batches": NB, "batch_note": f"step i searches batch i mod {NB}: {NB} distinct {Q}-query batches from the same generator, recall checked on each",
                   "replayed_batch_qps": world * Q * args.steps / wall_replayed, "replayed_single_stream_ms_per_launch": kernel_ms_replayed,
                   "ef_sweep": sweep, "shards": world,
                   "unit_note": f"one unit = one query searched against one {N}-vector shard",
                   "merged_queries_per_s": Q * args.steps / wall, "build_seconds": round(build_s, 1),
                   "exact_knn_seconds": round(exact_s, 3), "arith": "fast", "streams": n_streams,
                   "single_stream_ms_per_launch": kernel_ms},
    }

    if rank == 0:
        # ---- algorithmic bytes from the oracle's work counters on the SAME graph and queries ----
        from oracle import pyoracle
        ora = pyoracle.load()
        t0 = time.time()
        g = index.export_graph()
        ho = ora.HNSWIndex(ora.DistanceMetric(METRIC), M, EF_BUILD, 16, D)
        ho.import_graph(g)
        log(f"oracle: graph handed over in {time.time()-t0:.1f}s")
        with_cpu = world == 1 and not args.no_cpu_baseline          # the CPU baseline is timed at N = 1 only
        nqc = min(Q, args.cpu_queries or (Q if args.config == "c2" else 1000)) if with_cpu else Q
        ora.set_threads(1 if with_cpu else min(os.cpu_count() or 1, 16))
        t0 = time.perf_counter()
        ref, refn, rst = ho.search_batch(queries[:nqc], K, ef_search=ef, with_stats=True)
        cpu_s = time.perf_counter() - t0
        bytes_of = lambda c: c["rows_scored"].astype(np.float64) * (4 * info["row_stride_floats"]) + c["adj_entries"].astype(np.float64) * 4 + c["expanded"].astype(np.float64) * 8
        bytes_q = bytes_of(rst)
        if NB > 1:                                              # the sequential algorithm's work over every batch of the rotation
            ora.set_threads(min(os.cpu_count() or 1, 16))
            bytes_q = np.concatenate([bytes_q] + [bytes_of(ho.search_batch(batches[b][:nqc], K, ef_search=ef, with_stats=True)[2]) for b in range(1, NB)])
        alg_bytes = float(bytes_q.mean()) * Q
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of tools/collect_profiles.sh: reported only when they were measured on
        # this very kernel code (hash of its sources), at this ef and on this workload size
        traffic, traffic_note = None, "no counter measurement of this kernel code in profiles/pmc_traffic.json"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json" if args.config == "c2" else f"pmc_traffic_{args.config}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_sources_sha256_16") == kernel_sources_hash() and tj.get("ef_search") == ef and tj.get("n") == N and tj.get("queries") == Q:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_note = f"{tj.get('kernel')}, measured by {tj.get('command')}"
            except Exception:
                traffic = None
        achieved_overlapped = alg_bytes / (ms_per_step * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "traffic": traffic, "traffic_source": traffic_note, "kernel": "search_kernel", "kernel_ms": kernel_ms,
                              # the timed steps rotate over `streams` HIP streams, so consecutive launches overlap: the same
                              # algorithmic bytes over the time one batch takes in that steady state
                              "achieved_overlapped": achieved_overlapped, "frac_overlapped": achieved_overlapped / HBM_PEAK_GBS,
                              "ms_per_step_overlapped": ms_per_step, "streams": n_streams,
                              "traffic_frac_of_peak_overlapped": (traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                              "kernel_ms_replayed_batch": kernel_ms_replayed,
                              "algorithmic_bytes_per_query": float(bytes_q.mean()), "rows_scored_per_query": float(rst["rows_scored"].mean()),
                              "gpu_rows_scored_per_query": float(st["rows_scored"].mean())}
        # parity on the bench workload itself: same graph, same queries, same ef
        agree = float(np.mean(ref["id"][:, :K] == out["id"][:nqc]))
        result["config"]["id_agreement_with_oracle"] = round(agree, 5)          # this rank's shard, before the exchange
        if world == 1:
            result["config"]["oracle_recall_at_10"] = round(recall_at_k(ref, refn, exact[:nqc], K), 4)
            result["config"]["oracle_recall_note"] = "the oracle's sequential search on the graph this run built on the GPU (same graph, same queries, same ef)"
            if args.config == "c5":
                result["config"]["recall_note"] = ("recall of a 12.5M-row shard levels off below 0.95 with the reference's linking rule (M nearest, no diversity rule): "
                                                   "at 4M x 96 the oracle's point-by-point build and the batched GPU build follow the same curve within 0.5 pt "
                                                   "(profiles/r03_c5_plateau_4m_96_oracle.log, profiles/r03_c5_plateau_4m_96_gpu.log)")
        if with_cpu:
            cores = os.cpu_count() or 1
            result["cpu_baseline"] = {"value": nqc / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port",
                                      "sample": f"{nqc} of the {Q} queries, same graph, same ef={ef}, one query per call on one thread "
                                                f"(the reference's execution model); flat-array C++ port of hnsw.rs:513-554 "
                                                f"(dense arrays where the reference keeps HashMap<u64, Vec<f32>>: a generous baseline)"}
            ora.set_threads(min(cores, 16))
            t0 = time.perf_counter()
            ho.search_batch(queries[:nqc], K, ef_search=ef)
            result["cpu_baseline"]["value_all_cores"] = nqc / (time.perf_counter() - t0)
            result["cpu_baseline"]["cores_all"] = min(cores, 16)
            log(f"cpu baseline: {result['cpu_baseline']['value']:.0f} q/s on 1 thread, {result['cpu_baseline']['value_all_cores']:.0f} q/s on {min(cores,16)} threads")
        log(f"roofline: {achieved:.0f} GB/s algorithmic = {100*achieved/HBM_PEAK_GBS:.1f}% of {HBM_PEAK_GBS:.0f} GB/s; kernel {kernel_ms:.3f} ms/launch")

    if rank == 0:
        print(json.dumps(result), flush=True)
    if exchange:
        dist.destroy_process_group()


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X fp32 matrix peak (v_mfma_f32_32x32x2_f32), data sheet
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16)


def bench_c4(args):
    """10M x 768 dot product with payload filters: payload {cat: i mod 100, tier: mix(i) mod 10}; filters at 1 %, 5 %,
    10 % and 50 % selectivity compiled to device bitmaps; 10k-query batches on device buffers.  Narrow filters (at most
    1/8 of the rows pass) scan the passing rows in place (fp32 MFMA contraction + fused top-k + strict re-rank, rows read
    through a row map): bound by the matrix pipes, 2 Q N_pass D flop per batch.  Wide filters walk the graph with the pass
    bitmap as mask: bound by HBM, bytes from the kernel's own work counters (the oracle cannot build a graph this large)."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import vectordb_amd
    vdb = vectordb_amd.api()
    N, D, Q, K = args.n or 10_000_000, args.dim or 768, args.queries or 10_000, args.k
    EF = args.ef or 128
    t0 = time.time()
    centres = np.random.default_rng(0xC4).uniform(0, 128, size=(1024, D)).astype(np.float32)
    seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric.Dot, 16, 64, 16, D, seed=0xC4, filter_mode=vdb.FilterMode.Masked))
    ids_all = np.arange(N, dtype=np.uint64)
    cat = (ids_all % 100).astype(np.int64)
    tier = (((ids_all * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(10)).astype(np.int64)
    CH = 250_000
    sample_rows = {}                                             # rows kept on the host for the recall check
    for lo in range(0, N, CH):
        hi = min(N, lo + CH)
        blk = clustered(hi - lo, D, 0xC4 + 1 + lo // CH, centres)
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)         # unit rows (embedding-like)
        got = seg.insert_batch_columns(blk, {"cat": cat[lo:hi], "tier": tier[lo:hi]})
        assert int(got[0]) == lo + 1                             # segment ids start at 1 (segment.rs:27)
        keep = np.nonzero((cat[lo:hi] == 7) | (tier[lo:hi] == 3))[0]
        sample_rows[lo] = (keep + lo, blk[keep].copy())
        log(f"  c4: {hi} / {N} points ({time.time()-t0:.0f}s)")
    build_s = time.time() - t0
    queries = clustered(Q, D, 0xC4 + 5000, centres); queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    kept_ids = np.concatenate([v[0] for v in sample_rows.values()]); kept = np.concatenate([v[1] for v in sample_rows.values()])
    PV, F, Op = vdb.PayloadValue, vdb.Filter, vdb.ScalarComparisonOp
    cases = [("cat == 7 (1 %)", F.Match("cat", PV.Int(7)), cat == 7),
             ("tier == 3 and cat < 50 (5 %)", F.And([F.Match("tier", PV.Int(3)), F.Compare("cat", Op.Lt, PV.Int(50))]), (tier == 3) & (cat < 50)),
             ("tier == 3 (10 %)", F.Match("tier", PV.Int(3)), tier == 3),
             ("cat < 50 (50 %)", F.Compare("cat", Op.Lt, PV.Int(50)), cat < 50)]
    stream = torch.cuda.Stream(device=dev); sp = stream.cuda_stream
    d_q = torch.from_numpy(queries).to(dev)
    d_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev); d_n = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_st = torch.zeros(Q * 4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    info = seg.hnsw().device_info()
    results = []
    SAMPLE = 200                                                  # queries of the recall check (float64 brute force on the host)
    for name, f, mask in cases:
        t0 = time.perf_counter()
        n_pass, n_err = seg.compile_filter(f)
        compile_ms = (time.perf_counter() - t0) * 1e3
        call = lambda: vdb.check(vdb.fn.segment_search_batch_dev(seg._h, d_q.data_ptr(), Q, K, EF, f._h, d_out.data_ptr(), d_n.data_ptr(), d_st.data_ptr(), sp))
        for _ in range(max(1, args.warmup)): call()
        stream.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps): call()
        e1.record(stream); stream.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        out = d_out.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K); cnt = d_n.cpu().numpy().view(np.uint32)
        st = d_st.cpu().numpy().view(vdb.STATS_DTYPE).reshape(Q)
        rows = out["id"].astype(np.int64) - 1
        all_pass = bool(all(mask[rows[i][: cnt[i]]].all() for i in range(0, Q, 37)))
        rec = None
        sel = mask[kept_ids]
        if sel.sum() == mask.sum():                              # the passing rows are among those kept on the host
            sub = kept[sel].astype(np.float64); sub_ids = kept_ids[sel]
            top = np.argpartition(-(queries[:SAMPLE].astype(np.float64) @ sub.T), K, axis=1)[:, :K]
            rec = float(np.mean([len(set(rows[i][: cnt[i]].tolist()) & set(sub_ids[top[i]].tolist())) / K for i in range(SAMPLE)]))
        scanned = int(st["rows_scored"].sum()) == 0
        r = {"filter": name, "passing_rows": int(n_pass), "compile_ms_first_call": round(compile_ms, 2), "ms_per_batch": ms, "queries_per_s": Q / (ms * 1e-3),
             "path": "scan of the passing rows in place (bf16-split matrix-core candidate pass + strict fp32 re-rank)" if scanned else "masked graph walk", "mean_results": float(cnt.mean()),
             "all_results_pass_filter": all_pass, "recall_at_10_vs_float64_brute_force": rec}
        if scanned:
            tf = 2.0 * Q * n_pass * D / (ms * 1e-3) / 1e12
            # the candidate pass issues three bf16 MFMAs per fp32 product: priced against the dense bf16 matrix peak;
            # the algorithmic rate (2 Q N D) is kept beside it
            r["roofline"] = {"bound": "mfma", "achieved": 3 * tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 3 * tf / MFMA_BF16_PEAK_TFLOPS,
                             "peak_is": "dense bf16 MFMA; achieved = 3 bf16 products issued per fp32 product (bf16 split)",
                             "fp32_equivalent_tflops": tf,
                             "hbm_bytes_algorithmic": 4.0 * n_pass * D + N / 8}
        else:
            b = float(st["rows_scored"].astype(np.float64).sum() * (4 * info["row_stride_floats"] + 0.125) + st["adj_entries"].astype(np.float64).sum() * 4 + st["expanded"].astype(np.float64).sum() * 8)
            gbs = b / (ms * 1e-3) / 1e9
            r["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "rows_scored_per_query": float(st["rows_scored"].mean()), "bytes_from": "the kernel's own work counters"}
        results.append(r)
        log(f"  c4 {name}: {r['path']}, {ms:.2f} ms per {Q} queries = {r['queries_per_s']:.0f} q/s, roofline {r['roofline']['frac']:.3f} ({r['roofline']['bound']}), recall {rec}")
    head = results[2]
    line = {"metric": "queries/sec, 10M x 768 dot product with a payload filter (10 % of the rows pass)", "value": head["queries_per_s"], "unit": "queries/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_batch"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"embedding-shape synthetic: {N} x {D} fp32 dot product (unit rows, 1024 Gaussian centres), payload {{cat: i mod 100, tier: mix(i) mod 10}}, "
                                   f"{Q}-query device-buffer batches, HNSW M=16 ef_construction=64 ef_search={EF}, top-{K}, masked filter mode",
                       "filters": results, "segment_build_seconds": round(build_s, 1), "device_bytes": info["bytes_total"]},
            "roofline": head["roofline"]}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
