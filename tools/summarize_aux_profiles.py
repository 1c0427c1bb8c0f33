"""Turns the raw rocprofv3 output of tools/collect_aux_profiles.sh (gpurun_out/aux_<tag>/) into the committed
summaries profiles/<tag>_exact_knn.json, profiles/<tag>_scan.json and the kernel-stats tables beside them."""
import glob, json, os, re, sys
import pandas as pd

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = f"gpurun_out/aux_{tag}"
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
last_json = lambda p: json.loads([l for l in open(p).read().strip().splitlines() if l.startswith("{")][-1])

def stats(which, dst):
    df = pd.read_csv(newest(f"{src}/{which}/*/*_kernel_stats.csv"))
    df.to_csv(dst, index=False)
    return df

def counters(which, kernel_part):
    df = pd.read_csv(newest(f"{src}/{which}/*/*_counter_collection.csv"))
    k = df[df.Kernel_Name.str.contains(kernel_part, regex=False)]
    last = sorted(k.Dispatch_Id.unique())[-2:]
    g = k[k.Dispatch_Id.isin(last)].groupby(["Dispatch_Id", "Counter_Name"]).Counter_Value.sum().groupby("Counter_Name").mean()
    return {n: float(v) for n, v in g.items()}

def kernel_us(df, part):
    r = df[df.Name.str.contains(part, regex=False)]
    return {row.Name.split("(")[0][-60:]: {"calls": int(row.Calls), "avg_us": row.AverageNs / 1e3} for row in r.itertuples()}

# exact k-NN
ks = stats("knn_trace", f"profiles/{tag}_exact_knn_128_kernel_stats.csv")
ks7 = stats("knn768_trace", f"profiles/{tag}_exact_knn_768_kernel_stats.csv")
b = last_json(f"{src}/knn_trace.json"); b7 = last_json(f"{src}/knn768_trace.json")
cand = [n for n in ks.Name if "knn_bf16_kernel" in n or "knn_mfma_kernel" in n][0].split("(")[0].split("::")[-1].split("<")[0]   # the candidate-pass kernel of this build
sq = counters("knn_sq", cand); gr = counters("knn_grbm", cand)
mf = ks[ks.Name.str.contains(cand, regex=False)].iloc[0]
mfma_us = mf.AverageNs / 1e3
flop = 2.0 * 10000 * 1_000_000 * 128
clock = gr["GRBM_GUI_ACTIVE"] / 8 / mfma_us / 1e3
knn = {"command": f"tools/collect_aux_profiles.sh {tag}", "workload_128": b, "workload_768": b7,
       "kernels_128_us": kernel_us(ks, "knn_"), "kernels_768_us": kernel_us(ks7, "knn_"),
       "candidate_pass_128": {"kernel": mf.Name.split("(")[0], "avg_us": mfma_us, "tflops": flop / mfma_us / 1e6, "frac_of_157.3": flop / mfma_us / 1e6 / 157.3,
                              "counters_per_launch": sq, "clock_GHz": clock,
                              "mfma_busy_share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * mfma_us * 1e3 * clock),
                              "non_mfma_valu_per_mfma": (sq["SQ_INSTS_VALU"] - sq["SQ_INSTS_MFMA"]) / sq["SQ_INSTS_MFMA"]}}
json.dump(knn, open(f"profiles/{tag}_exact_knn.json", "w"), indent=1)

# scan
ss = stats("scan_trace", f"profiles/{tag}_scan_960_kernel_stats.csv")
s8 = last_json(f"{src}/scan_trace.json"); s1 = last_json(f"{src}/scan1_trace.json")
ss1 = pd.read_csv(newest(f"{src}/scan1_trace/*/*_kernel_stats.csv"))
f = counters("scan_fetch", "scan_"); w = counters("scan_write", "scan_")
k8 = ss[ss.Name.str.contains("scan_", regex=False)].iloc[0]; k1 = ss1[ss1.Name.str.contains("scan_", regex=False)].iloc[0]
hbm = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024
scan = {"command": f"tools/collect_aux_profiles.sh {tag}", "eight_queries": s8, "one_query": s1,
        "kernel": re.search(r"scan_\w+<[^>]*>", k8.Name).group(0), "avg_us_8_queries": k8.AverageNs / 1e3, "avg_us_1_query": k1.AverageNs / 1e3,
        "algorithmic_GBps_8_queries": s8["algorithmic_bytes"] / (k8.AverageNs / 1e3) / 1e3, "algorithmic_GBps_1_query": s1["algorithmic_bytes"] / (k1.AverageNs / 1e3) / 1e3,
        "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
        "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane read, so it is doubled; WRITE_SIZE is taken as is",
        "hbm_bytes_per_launch_8_queries": hbm, "traffic_over_algorithmic": hbm / s8["algorithmic_bytes"]}
if glob.glob(f"{src}/scan_l2_trace/*/*_kernel_stats.csv"):
    sl = pd.read_csv(newest(f"{src}/scan_l2_trace/*/*_kernel_stats.csv")); kl = sl[sl.Name.str.contains("scan_", regex=False)].iloc[0]; bl = last_json(f"{src}/scan_l2_trace.json")
    scan["eight_queries_l2"] = {"kernel": re.search(r"scan_\w+<[^>]*>", kl.Name).group(0), "avg_us": kl.AverageNs / 1e3, "bench": bl, "frac_of_8TBps": bl["algorithmic_bytes"] / (kl.AverageNs / 1e3) / 1e3 / 8000}
scan["kernel_1_query"] = re.search(r"scan_\w+<[^>]*>", k1.Name).group(0)
scan["frac_of_8TBps_8_queries"] = scan["algorithmic_GBps_8_queries"] / 8000; scan["frac_of_8TBps_1_query"] = scan["algorithmic_GBps_1_query"] / 8000
json.dump(scan, open(f"profiles/{tag}_scan.json", "w"), indent=1)
print(json.dumps(knn["candidate_pass_128"], indent=1)); print(json.dumps({k: v for k, v in scan.items() if not isinstance(v, dict)}, indent=1))
