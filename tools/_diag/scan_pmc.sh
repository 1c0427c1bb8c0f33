export TMPDIR=/tmp
OUT=gpurun_out/scan_pmc; rm -rf $OUT; mkdir -p $OUT
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
VDB_SCAN_NO_MFMA=1 timeout -s KILL 100 rocprofv3 --pmc $C --output-format csv -d $OUT/old8 -- python3 tools/bench_scan.py --dim 960 --steps 3 > $OUT/old8.log 2>&1 && echo old8 ok
timeout -s KILL 100 rocprofv3 --pmc $C --output-format csv -d $OUT/old1 -- python3 tools/bench_scan.py --dim 960 --steps 3 --queries 1 > $OUT/old1.log 2>&1 && echo old1 ok
VDB_SCAN_GC=12 timeout -s KILL 100 rocprofv3 --pmc $C --output-format csv -d $OUT/mfma -- python3 tools/bench_scan.py --dim 960 --steps 3 > $OUT/mfma.log 2>&1 && echo mfma ok
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
VDB_SCAN_NO_MFMA=1 timeout -s KILL 100 rocprofv3 --pmc $C2 --output-format csv -d $OUT/old8b -- python3 tools/bench_scan.py --dim 960 --steps 3 > $OUT/old8b.log 2>&1 && echo old8b ok
timeout -s KILL 100 rocprofv3 --pmc $C2 --output-format csv -d $OUT/old1b -- python3 tools/bench_scan.py --dim 960 --steps 3 --queries 1 > $OUT/old1b.log 2>&1 && echo old1b ok
python3 - <<'PY'
import csv, glob, collections
for tag in ("old8","old1","mfma","old8b","old1b"):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob(f"gpurun_out/scan_pmc/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "scan_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    print(tag, {k: round(v/max(cnt[k],1)) for k,v in acc.items()}, dict(cnt).get("SQ_WAVE_CYCLES"))
PY
