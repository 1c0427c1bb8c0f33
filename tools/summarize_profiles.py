"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed
summaries under profiles/: kernel stats, per-dispatch table of the timed kernel, HBM traffic per launch."""
import glob, json, os, sys
import pandas as pd

def newest(pattern):          # earlier collections may still lie in the same directory
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
bench = json.loads(open(f"{src}/trace.json").read().strip().splitlines()[-1])
ef = bench["config"]["ef_search"]
stats = pd.read_csv(newest(f"{src}/trace/*/*_kernel_stats.csv"))
stats.to_csv(f"profiles/{tag}_bench_ef{ef}_kernel_stats.csv", index=False)
tr = pd.read_csv(newest(f"{src}/trace/*/*_kernel_trace.csv"))
sk = tr[tr.Kernel_Name.str.contains(r"search_kernel<0")].copy()
sk["Duration_us"] = (sk.End_Timestamp - sk.Start_Timestamp) / 1e3
main_name = sk.groupby("Kernel_Name").Duration_us.sum().idxmax()
timed = sk[sk.Kernel_Name == main_name].tail(bench["steps"])
cols = ["Kernel_Name", "Duration_us", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"]
sk[[c for c in cols if c in sk.columns]].tail(3 * bench["steps"]).to_csv(f"profiles/{tag}_bench_ef{ef}_search_kernel_dispatches.csv", index=False)

def pmc(which):
    df = pd.read_csv(newest(f"{src}/{which}/*/*_counter_collection.csv"))
    k = df[df.Kernel_Name == main_name]
    last = sorted(k.Dispatch_Id.unique())[-5:]
    return float(k[k.Dispatch_Id.isin(last)].groupby("Dispatch_Id").Counter_Value.sum().mean()), len(last)

fetch_kb, n = pmc("fetch"); write_kb, _ = pmc("write")
out = {
    "command": f"tools/collect_profiles.sh {tag} {ef}  (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over: python3 bench.py --ef {ef} --streams 1 --no-cpu-baseline --steps 5 --warmup 1)",
    "kernel": main_name, "ef_search": ef, "launches_averaged": n,
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane read, so it is doubled; WRITE_SIZE is taken as is",
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "kernel_avg_us_rocprof": float(timed.Duration_us.mean()), "kernel_ms_bench_events": bench["roofline"]["kernel_ms"] if "roofline" in bench else bench["config"]["single_stream_ms_per_launch"],
    "followup_launch_avg_us": float(sk[sk.Kernel_Name != main_name].Duration_us.mean()) if (sk.Kernel_Name != main_name).any() else 0.0,
    "round": 1,
}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
