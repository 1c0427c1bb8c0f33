#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel traces of the headline bench (one stream: lone launches;
# default streams: overlapping launches) and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters, GRBM).
# Raw output lands under gpurun_out/; tools/summarize_profiles.py reduces it to profiles/.
# usage: tools/collect_profiles.sh <tag> <ef> [config] [passes]   (config: c2 (default), c3, c5; passes: "all" or "traffic" = traces + FETCH / WRITE only)
set -e
TAG=${1:-r03}; EF=${2:-0}; CFG=${3:-c2}; PASSES=${4:-all}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_$CFG; rm -rf $OUT; mkdir -p $OUT
(timeout 60 rocprofv3 -L > $OUT/counters.txt 2>&1 || true)
LONE="--config $CFG --ef $EF --streams 1 --no-cpu-baseline"
BOTH="--config $CFG --ef $EF --no-cpu-baseline"
T=${VDB_PROFILE_TIMEOUT:-300}
timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $LONE --steps 20 --warmup 3 > $OUT/trace.json 2> $OUT/trace.log
echo trace done
timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py $BOTH --steps 20 --warmup 3 > $OUT/trace2.json 2> $OUT/trace2.log
echo trace2 done
SQ_PASSES=("sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "grbm GRBM_GUI_ACTIVE")
if [ "$PASSES" != "all" ]; then SQ_PASSES=(); fi
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "${SQ_PASSES[@]}"; do
  set -- $pass; name=$1; shift
  timeout -s KILL $T rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $LONE --steps 5 --warmup 1 > $OUT/$name.json 2> $OUT/$name.log && echo $name done || echo $name FAILED
done
for pass in "fetch2 FETCH_SIZE" "write2 WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout -s KILL $T rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $BOTH --steps 6 --warmup 2 > $OUT/$name.json 2> $OUT/$name.log && echo $name done || echo $name FAILED
done
echo collected $OUT
