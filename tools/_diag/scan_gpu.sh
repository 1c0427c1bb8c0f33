set -e
echo "mfma small"; VDB_SCAN_MFMA_MIN=64 timeout -k 10 120 python tools/bench_scan.py --dim 64,96,128,192,256,384,512 --steps 20 2>/dev/null
echo "old small"; VDB_SCAN_NO_MFMA=1 timeout -k 10 120 python tools/bench_scan.py --dim 64,96,128,192,256,384,512 --steps 20 2>/dev/null
echo "mfma 3q/4q/5q 960 vs old"; for q in 3 4 5; do timeout -k 10 120 python tools/bench_scan.py --dim 960 --steps 20 --queries $q 2>/dev/null; VDB_SCAN_NO_MFMA=1 timeout -k 10 120 python tools/bench_scan.py --dim 960 --steps 20 --queries $q 2>/dev/null; done
