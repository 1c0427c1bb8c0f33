#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel traces of a bench configuration (one stream: lone launches;
# default streams: overlapping launches) and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters, GRBM).
# Raw output lands under gpurun_out/; tools/summarize_profiles.py reduces it to profiles/.
# usage: tools/collect_profiles.sh <tag> <ef> [config] [passes]
#   config: c2 (default), c3, c5;  passes: "all", "traffic" (traces + FETCH / WRITE) or a list of pass names, e.g. "trace fetch"
#   (a configuration whose set-up takes minutes is collected over several calls; the output directory is kept between them)
set -e
TAG=${1:-r03}; EF=${2:-0}; CFG=${3:-c2}; PASSES=${4:-all}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_$CFG; mkdir -p $OUT
LONE="--config $CFG --ef $EF --streams 1 --no-cpu-baseline"
BOTH="--config $CFG --ef $EF --no-cpu-baseline"
T=${VDB_PROFILE_TIMEOUT:-300}
case "$PASSES" in
  all) LIST="trace trace2 fetch write sq sq2 grbm fetch2 write2";;
  traffic) LIST="trace trace2 fetch write fetch2 write2";;
  *) LIST="$PASSES";;
esac
counters() {
  case "$1" in
    fetch|fetch2) echo FETCH_SIZE;; write|write2) echo WRITE_SIZE;;
    sq) echo SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY;;
    sq2) echo SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;;
    grbm) echo GRBM_GUI_ACTIVE;;
  esac
}
for name in $LIST; do
  rm -rf $OUT/$name
  case "$name" in
    trace)  timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $LONE --steps 20 --warmup 3 > $OUT/trace.json 2> $OUT/trace.log;;
    trace2) timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py $BOTH --steps 20 --warmup 3 > $OUT/trace2.json 2> $OUT/trace2.log;;
    fetch2|write2) timeout -s KILL $T rocprofv3 --pmc $(counters $name) --output-format csv -d $OUT/$name -- python3 bench.py $BOTH --steps 6 --warmup 2 > $OUT/$name.json 2> $OUT/$name.log;;
    *) timeout -s KILL $T rocprofv3 --pmc $(counters $name) --output-format csv -d $OUT/$name -- python3 bench.py $LONE --steps 5 --warmup 1 > $OUT/$name.json 2> $OUT/$name.log;;
  esac
  echo $name done
done
echo collected $OUT
