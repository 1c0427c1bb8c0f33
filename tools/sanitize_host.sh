#!/bin/bash
# Host-side sanitizer runs (CPU only, no GPU code is instrumented):
#   oracle   the CPU restatement built with AddressSanitizer + UndefinedBehaviorSanitizer (g++), CPU test suite on it
#   library  the host side of libvdbmi built with AddressSanitizer + UBSan (hipcc -Xarch_host), CPU test suite on it
# usage: tools/sanitize_host.sh oracle|library     (log: /tmp/sanitize_<which>.log; exit status = pytest's)
set -e
cd "$(dirname "$0")/.."
which=${1:-oracle}
if [ "$which" = oracle ]; then
  mkdir -p oracle/_build
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
      -pthread -shared -Wl,-Bsymbolic -o oracle/_build/libvdb_oracle_san.so oracle/ora_capi.cpp
  RT=$(gcc -print-file-name=libasan.so)
  LD_PRELOAD="$RT:$(gcc -print-file-name=libstdc++.so.6)${LD_PRELOAD:+:$LD_PRELOAD}" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    VDB_ORACLE_LIBRARY=$PWD/oracle/_build/libvdb_oracle_san.so python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/sanitize_oracle.log
  exit ${PIPESTATUS[0]}
else
  B=vectordb_amd/_csrc_san; rm -rf $B; mkdir -p $B tools/_diag
  cp vectordb_amd/csrc/*.hip vectordb_amd/csrc/*.cpp vectordb_amd/csrc/*.hpp vectordb_amd/csrc/Makefile $B/
  SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
  make -C $B -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed $SAN" index.o segment.o capi.o >/dev/null
  for o in kernels knn_mfma filter_kernels search_plain search_build search_masked search_literal; do cp vectordb_amd/csrc/$o.o $B/; done
  RTDIR=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -shared-libsan -o tools/_diag/libvdbmi_san.so $B/*.o
  rm -rf $B
  LD_PRELOAD="$RTDIR/libclang_rt.asan-x86_64.so${LD_PRELOAD:+:$LD_PRELOAD}" LD_LIBRARY_PATH="$RTDIR:$LD_LIBRARY_PATH" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1 VDB_LIBRARY=$PWD/tools/_diag/libvdbmi_san.so python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/sanitize_library.log
  exit ${PIPESTATUS[0]}
fi
