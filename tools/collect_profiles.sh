#!/bin/bash
# Runs on the GPU box (through gpurun): the headline bench, a rocprofv3 kernel trace of the same
# command, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).  Raw output lands under gpurun_out/.
# usage: tools/collect_profiles.sh <tag> <ef>
set -e
TAG=${1:-r01}; EF=${2:-0}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
ARGS="--ef $EF --streams 1 --no-cpu-baseline --steps 20 --warmup 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --ef $EF --streams 1 --no-cpu-baseline --steps 5 --warmup 1 > $OUT/fetch.json 2> $OUT/fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --ef $EF --streams 1 --no-cpu-baseline --steps 5 --warmup 1 > $OUT/write.json 2> $OUT/write.log
echo collected $OUT
