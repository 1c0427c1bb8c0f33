"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed
summaries under profiles/: kernel stats of the lone-launch and the overlapped (multi-stream) run, the timed dispatches
of the overlapped run with their start / end times, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes
(stamped with a hash of the kernel's sources, which bench.py checks before it reports the figure) and the SQ / GRBM
counters of the timed kernel."""
import glob, json, os, sys
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_hash

def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

# usage: summarize_profiles.py <tag> [config]
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
src = f"gpurun_out/prof_{tag}_{cfg}"
os.makedirs("profiles", exist_ok=True)
last_json = lambda p: json.loads(open(p).read().strip().splitlines()[-1])
bench = last_json(f"{src}/trace.json")
have2 = os.path.exists(f"{src}/trace2.json") and os.path.exists(f"{src}/fetch2.json")     # the overlapped passes may be left out
bench2 = last_json(f"{src}/trace2.json") if have2 else bench
ef = bench["config"]["ef_search"]
pre = f"profiles/{tag}_bench_ef{ef}" if cfg == "c2" else f"profiles/{tag}_bench_{cfg}_ef{ef}"

def trace(which, b):
    stats = pd.read_csv(newest(f"{src}/{which}/*/*_kernel_stats.csv"))
    tr = pd.read_csv(newest(f"{src}/{which}/*/*_kernel_trace.csv"))
    sk = tr[tr.Kernel_Name.str.contains(r"search_kernel<0|search_team_kernel<")].copy()
    sk["Duration_us"] = (sk.End_Timestamp - sk.Start_Timestamp) / 1e3
    main = sk.groupby("Kernel_Name").Duration_us.sum().idxmax()
    timed = sk[sk.Kernel_Name == main].sort_values("Start_Timestamp").tail(b["steps"]).copy()
    return stats, sk, main, timed

stats, sk, main_name, timed = trace("trace", bench)
stats.to_csv(f"{pre}_lone_kernel_stats.csv", index=False)
stats2, sk2, main2, timed2 = trace("trace2" if have2 else "trace", bench2)
stats2.to_csv(f"{pre}_overlapped_kernel_stats.csv", index=False)
assert main2 == main_name
t0 = timed2.Start_Timestamp.min()
timed2["start_us"] = (timed2.Start_Timestamp - t0) / 1e3; timed2["end_us"] = (timed2.End_Timestamp - t0) / 1e3
cols = ["Kernel_Name", "Queue_Id", "Stream_Id", "start_us", "end_us", "Duration_us", "Grid_Size_X", "Workgroup_Size_X"]
timed2[[c for c in cols if c in timed2.columns]].to_csv(f"{pre}_overlapped_dispatches.csv", index=False)
span = timed2.end_us.max() - timed2.start_us.min()
ends = sorted(timed2.end_us)
period = (ends[-1] - ends[len(ends) // 2]) / (len(ends) - 1 - len(ends) // 2)      # steady state: spacing of completions
both = 0.0                                                                            # time with two launches running
ev = sorted([(s, 1) for s in timed2.start_us] + [(e, -1) for e in timed2.end_us]); live = 0; prev = ev[0][0]
for t, d in ev:
    if live >= 2: both += t - prev
    live += d; prev = t

def pmc(which, names=None):
    df = pd.read_csv(newest(f"{src}/{which}/*/*_counter_collection.csv"))
    k = df[df.Kernel_Name == main_name]
    last = sorted(k.Dispatch_Id.unique())[-4:]          # timed launches (the last ones of the run)
    g = k[k.Dispatch_Id.isin(last)].groupby(["Dispatch_Id", "Counter_Name"]).Counter_Value.sum().groupby("Counter_Name").mean()
    return {n: float(v) for n, v in g.items()}, len(last)

f1, n1 = pmc("fetch"); w1, _ = pmc("write"); f2, _ = pmc("fetch2" if have2 else "fetch"); w2, _ = pmc("write2" if have2 else "write")
hbm1 = (2 * f1["FETCH_SIZE"] + w1["WRITE_SIZE"]) * 1024; hbm2 = (2 * f2["FETCH_SIZE"] + w2["WRITE_SIZE"]) * 1024
lone_us = float(timed.Duration_us.mean())
N = int(bench["config"]["total_rows"]) // max(1, int(bench["config"]["shards"])); Q = int(bench["config"]["workload"].split("fp32")[1].split("-query")[0].split(",")[-1].strip())
out = {
    "command": f"tools/collect_profiles.sh {tag} {ef} {cfg}  (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each, over python3 bench.py --config {cfg} --ef {ef} --no-cpu-baseline with --streams 1 and with the default streams; the timed steps rotate through {bench['config'].get('query_batches', 1)} distinct query batches)",
    "kernel": main_name, "kernel_sources_sha256_16": kernel_sources_hash(), "ef_search": ef, "n": N, "queries": Q, "launches_averaged": n1,
    "FETCH_SIZE_KB": f1["FETCH_SIZE"], "WRITE_SIZE_KB": w1["WRITE_SIZE"],
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane read, so it is doubled; WRITE_SIZE is taken as is",
    "hbm_bytes_per_launch": hbm1,
    "kernel_avg_us_rocprof": lone_us, "kernel_ms_bench_events": bench["config"]["single_stream_ms_per_launch"],
    "lone_launch": {"hbm_GBps": hbm1 / lone_us / 1e3, "frac_of_8TBps": hbm1 / lone_us / 1e3 / 8000},
    "overlapped": {"streams": bench2["config"]["streams"], "FETCH_SIZE_KB": f2["FETCH_SIZE"], "WRITE_SIZE_KB": w2["WRITE_SIZE"], "hbm_bytes_per_launch": hbm2,
                   "launches": int(len(timed2)), "span_us": float(span), "us_per_launch_steady_state": float(period),
                   "share_of_span_with_two_launches_running": both / span,
                   "bench_ms_per_step": bench2["ms_per_step"], "bench_value_qps": bench2["value"],
                   "hbm_GBps": hbm2 / (bench2["ms_per_step"] * 1e3) / 1e3, "frac_of_8TBps": hbm2 / (bench2["ms_per_step"] * 1e3) / 1e3 / 8000},
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_query"] * Q,
    "overlapped_passes_collected": have2,
    "round": 3,
}
out["traffic_over_algorithmic"] = hbm1 / out["algorithmic_bytes_per_launch"]
json.dump(out, open("profiles/pmc_traffic.json" if cfg == "c2" else f"profiles/pmc_traffic_{cfg}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
if not glob.glob(f"{src}/sq/*/*_counter_collection.csv"):
    sys.exit(0)
sq, _ = pmc("sq"); sq2, _ = pmc("sq2"); gr, _ = pmc("grbm")
waves = sq["SQ_WAVES"]
cnt = {"kernel": main_name, "kernel_sources_sha256_16": kernel_sources_hash(), "launch": f"{int(waves)} wavefronts (one per query), lone launches (--streams 1)",
       "counters_per_launch": {**sq, **sq2, **gr},
       "derived": {"valu_per_query": sq["SQ_INSTS_VALU"] / waves, "salu_per_query": sq["SQ_INSTS_SALU"] / waves, "lds_per_query": sq["SQ_INSTS_LDS"] / waves,
                   "vmem_rd_per_query": sq2.get("SQ_INSTS_VMEM_RD", 0) / waves,
                   "share_of_wave_cycles_waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                   "share_of_wave_cycles_issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": sq2["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                   "share_of_wave_cycles_issue_stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": sq2["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                   "clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)": gr["GRBM_GUI_ACTIVE"] / 8 / lone_us / 1e3,
                   "lds_bank_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": sq2["SQ_LDS_BANK_CONFLICT"] / sq2["SQ_LDS_IDX_ACTIVE"]}}
json.dump(cnt, open(f"profiles/{tag}_search_kernel_sq_counters.json" if cfg == "c2" else f"profiles/{tag}_search_kernel_{cfg}_sq_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)); print(json.dumps(cnt["derived"], indent=1))
