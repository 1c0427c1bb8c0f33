"""Times the streaming distance scan (vdb_score_scan_dev, FAST arithmetic) on device buffers: N x D fp32 rows against
one pass of 8 queries (the number the kernel holds in LDS).  Algorithmic bytes = 4 N stride read + 4 nq N written.
Prints one JSON line with GB/s against the 8 TB/s HBM peak."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--dim", type=str, default="960", help="comma-separated list")
ap.add_argument("--queries", type=int, default=8); ap.add_argument("--metric", type=int, default=0); ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
import torch
import vectordb_amd
vdb = vectordb_amd.api()
N, Q = args.n, args.queries
for D in [int(t) for t in args.dim.split(",")]:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    base = torch.rand((N, D), generator=g, device=dev, dtype=torch.float32); q = torch.rand((Q, D), generator=g, device=dev, dtype=torch.float32)
    out = torch.zeros((Q, N), device=dev, dtype=torch.float32)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    run = lambda: vdb.check(vdb.fn.score_scan_dev(q.data_ptr(), Q, base.data_ptr(), N, D, D, args.metric, int(vdb.Arith.Fast), out.data_ptr(), st.cuda_stream))
    for _ in range(3): run()
    st.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.steps): run()
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    # spot check against torch
    ref = (1 - (q[:2] @ base[:1000].T) / (q[:2].norm(dim=1)[:, None] * base[:1000].norm(dim=1)[None, :] + 1e-10)) if args.metric == 0 else None
    err = float((out[:2, :1000] - ref).abs().max()) if ref is not None else None
    bytes_ = 4.0 * N * D + 4.0 * Q * N
    print(json.dumps({"workload": f"distance scan {Q} x {N} x {D}, metric {args.metric}, FAST", "ms_per_pass": ms, "algorithmic_bytes": bytes_,
                      "GBps": bytes_ / (ms * 1e-3) / 1e9, "frac_of_8TBps": bytes_ / (ms * 1e-3) / 1e9 / 8000, "max_abs_err_vs_torch_sample": err}))
