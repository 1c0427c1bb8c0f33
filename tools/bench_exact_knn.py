"""Times the exact k-NN path (10k queries x 1M x 128 by default) on one GPU: device-resident queries and
results, HIP events on the launch stream.  Prints one JSON line with seconds per batch and the
TFLOP/s of the contraction in fp32 terms (2*Q*N*D flop) against the fp32 MFMA peak; the candidate pass itself runs
three bf16 MFMAs per product (VDB_KNN_FP32=1: the fp32 MFMA kernels), so `frac` can exceed 1."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--queries", type=int, default=10_000); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--metric", type=int, default=2)
ap.add_argument("--graph", action="store_true", help="build the index's graph (the exact search then starts from the graph's answer); default: rows only")
args = ap.parse_args()
import torch
import vectordb_amd
vdb = vectordb_amd.api()
N, D, Q, K = args.n, args.dim, args.queries, args.k
centres = np.random.default_rng(0xC2).uniform(0, 128, size=(4096, D)).astype(np.float32)
base = clustered(N, D, 0xC3, centres); queries = clustered(Q, D, 0xC2 + 1000, centres)
g = dict(ids=np.arange(N, dtype=np.uint64), levels=np.zeros(N, np.uint32), vectors=base, deleted=np.zeros(N, np.uint8),
         layers=[], entry=0, current_max_level=0)
idx = vdb.HNSWIndex(vdb.DistanceMetric(args.metric), 16, 64, 16, D, arith=vdb.Arith.Fast)
if args.graph:
    for lo in range(0, N, 250_000): idx.insert_batch(g["ids"][lo:lo + 250_000], base[lo:lo + 250_000])
else:
    idx.import_graph(g)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
d_q = torch.from_numpy(queries).to(dev)
d_out = torch.zeros(Q * K * 16, dtype=torch.uint8, device=dev); d_n = torch.zeros(Q, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
def run():
    vdb.check(vdb.fn.index_exact_knn_batch_dev(idx._h, d_q.data_ptr(), Q, K, d_out.data_ptr(), d_n.data_ptr(), st.cuda_stream))
run(); st.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(args.steps): run()
e1.record(st); st.synchronize()
sec = e0.elapsed_time(e1) / 1e3 / args.steps
tf = 2.0 * Q * N * D / sec / 1e12
print(json.dumps({"workload": f"exact k-NN {Q} x {N} x {D}, top-{K}, metric {args.metric}", "index": "rows and graph (the candidate pass starts from the graph's answer)" if args.graph and not os.environ.get("VDB_KNN_NO_HINT") else "rows only (bounds start at +inf)", "seconds_per_batch": sec,
                  "tflops": tf, "peak_tflops_fp32_mfma": 157.3, "frac": tf / 157.3,
                  "candidate_pass": "fp32 MFMA" if os.environ.get("VDB_KNN_FP32") else "bf16 split, 3 MFMAs per product", "bf16_tflops_issued": None if os.environ.get("VDB_KNN_FP32") else 3 * tf, "peak_tflops_bf16_mfma": 2500.0}))
