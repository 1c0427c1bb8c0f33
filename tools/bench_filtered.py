"""Times Segment::search_with_filter in batches on one GPU (host buffers in and out, so the figure
includes the PCIe copies and the filter compilation): payload {cat: i mod 100, tier: mix(i) mod 10},
filters at 1 %, 10 % and 5 % selectivity, both filter modes.  Recall is against a brute-force scan
of the passing rows.  Prints one JSON line per (filter, mode)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clustered

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200_000); ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--queries", type=int, default=2000); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--ef", type=int, default=128); ap.add_argument("--metric", type=int, default=2)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--modes", default="masked,literal")
args = ap.parse_args()
import vectordb_amd
vdb = vectordb_amd.api()
N, D, Q, K = args.n, args.dim, args.queries, args.k
centres = np.random.default_rng(0xC4).uniform(0, 128, size=(1024, D)).astype(np.float32)
base = clustered(N, D, 0xC4 + 1, centres); queries = clustered(Q, D, 0xC4 + 2, centres)
if args.metric != 2:
    base /= np.linalg.norm(base, axis=1, keepdims=True); queries /= np.linalg.norm(queries, axis=1, keepdims=True)
ids = np.arange(N, dtype=np.uint64)
FIRST_ID = 1          # Segment ids start at 1 (segment.rs:27)
cat = (ids % 100).astype(np.int64)
tier = (((ids * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(10)).astype(np.int64)

print(f"[filtered] {N} x {D} rows generated", file=sys.stderr, flush=True)
t0 = time.perf_counter()
payloads = []
for i in range(N):
    p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(int(cat[i]))); p.set("tier", vdb.PayloadValue.Int(int(tier[i])))
    payloads.append(p)
t_payload = time.perf_counter() - t0
def build(mode):
    seg = vdb.Segment(vdb.HNSWIndex(vdb.DistanceMetric(args.metric), 16, 64, 16, D, seed=0xC4, filter_mode=mode))
    t0 = time.perf_counter()
    CH = 100_000
    got = []
    for lo in range(0, N, CH):
        got.append(seg.insert_batch(base[lo:lo + CH], payloads[lo:lo + CH]))
        print(f"[filtered] {mode.name}: {min(N, lo + CH)} points in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    assert np.array_equal(np.concatenate(got), ids + np.uint64(FIRST_ID)), "segment ids are expected to run from FIRST_ID"
    print(f"[filtered] {mode.name}: segment of {N} built in {time.perf_counter() - t0:.1f}s (payload objects: {t_payload:.1f}s)", file=sys.stderr)
    return seg

PV, F, Op = vdb.PayloadValue, vdb.Filter, vdb.ScalarComparisonOp
cases = [
    ("cat == 7 (1 %)", lambda: F.Match("cat", PV.Int(7)), cat == 7),
    ("tier == 3 (10 %)", lambda: F.Match("tier", PV.Int(3)), tier == 3),
    ("tier == 3 and cat < 50 (5 %)", lambda: F.And([F.Match("tier", PV.Int(3)), F.Compare("cat", Op.Lt, PV.Int(50))]), (tier == 3) & (cat < 50)),
    ("cat < 50 (50 %)", lambda: F.Compare("cat", Op.Lt, PV.Int(50)), cat < 50),
]

def ground_truth(mask):
    rows = np.nonzero(mask)[0]
    sub = base[rows].astype(np.float64); q = queries.astype(np.float64)
    if args.metric == 2: dist = (q * q).sum(1)[:, None] - 2.0 * q @ sub.T + (sub * sub).sum(1)[None, :]
    else: dist = -(q @ sub.T)
    top = np.argpartition(dist, K, axis=1)[:, :K]
    return [set(rows[t].tolist()) for t in top]

truths = []
for name, _, mask in cases:
    truths.append(ground_truth(mask))
    print(f"[filtered] ground truth for {name}: done", file=sys.stderr, flush=True)
for mode in [m for m in (vdb.FilterMode.Masked, vdb.FilterMode.Literal) if m.name.lower() in args.modes.split(",")]:
    seg = build(mode)
    for (name, make, mask), truth in zip(cases, truths):
        f = make()
        try:
            out, cnt = seg.search_batch(queries, K, ef_search=args.ef, filter=f)      # warm-up (bitmap cache, buffers)
        except vdb.DBError as e:
            print(json.dumps({"filter": name, "mode": mode.name, "error": str(e)})); continue
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, cnt = seg.search_batch(queries, K, ef_search=args.ef, filter=f)
        sec = (time.perf_counter() - t0) / args.steps
        rows = out["id"].astype(np.int64) - FIRST_ID
        hits = sum(len(set(rows[i][: cnt[i]].tolist()) & truth[i]) for i in range(Q))
        passing = int(mask.sum())
        ok = all(mask[rows[i][: cnt[i]]].all() for i in range(Q))
        print(json.dumps({"workload": f"filtered search {Q} x {N} x {D}, top-{K}, ef {args.ef}", "filter": name, "mode": mode.name,
                          "passing_rows": passing, "seconds_per_batch": sec, "queries_per_s": Q / sec,
                          "recall_at_k_vs_filtered_brute_force": hits / (Q * K), "mean_results": float(cnt.mean()),
                          "all_results_pass_filter": bool(ok)}))
