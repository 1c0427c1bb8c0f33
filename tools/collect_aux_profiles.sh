#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel traces and counter passes of the exact k-NN path
# (tools/bench_exact_knn.py: 10k x 1M x 128 and 10k x 1M x 768) and of the streaming distance scan
# (tools/bench_scan.py: 8 x 1M x 960).  Raw output under gpurun_out/; tools/summarize_aux_profiles.py reduces it.
# usage: tools/collect_aux_profiles.sh <tag> [all|knn|scan]   (the output directory is kept between calls)
TAG=${1:-r03}; WHAT=${2:-all}
export TMPDIR=/tmp
OUT=gpurun_out/aux_$TAG; mkdir -p $OUT
run() { name=$1; shift; rm -rf $OUT/$name; timeout -s KILL 200 rocprofv3 "$@" > $OUT/$name.json 2> $OUT/$name.log && echo $name done || { echo $name FAILED; tail -3 $OUT/$name.log; }; }
if [ $WHAT != scan ]; then
run knn_trace --kernel-trace --stats --output-format csv -d $OUT/knn_trace -- python3 tools/bench_exact_knn.py --graph --steps 5
run knn768_trace --kernel-trace --stats --output-format csv -d $OUT/knn768_trace -- python3 tools/bench_exact_knn.py --dim 768 --metric 1 --steps 2
run knn_sq --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/knn_sq -- python3 tools/bench_exact_knn.py --graph --steps 2
run knn_sq2 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/knn_sq2 -- python3 tools/bench_exact_knn.py --graph --steps 2
run knn_grbm --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/knn_grbm -- python3 tools/bench_exact_knn.py --graph --steps 2
fi
if [ $WHAT != knn ]; then
run scan_trace --kernel-trace --stats --output-format csv -d $OUT/scan_trace -- python3 tools/bench_scan.py --dim 960 --steps 20
run scan1_trace --kernel-trace --stats --output-format csv -d $OUT/scan1_trace -- python3 tools/bench_scan.py --dim 960 --queries 1 --steps 20
run scan_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/scan_fetch -- python3 tools/bench_scan.py --dim 960 --steps 5
run scan_write --pmc WRITE_SIZE --output-format csv -d $OUT/scan_write -- python3 tools/bench_scan.py --dim 960 --steps 5
run scan_l2_trace --kernel-trace --stats --output-format csv -d $OUT/scan_l2_trace -- python3 tools/bench_scan.py --dim 960 --metric 2 --steps 20
fi
echo collected $OUT
