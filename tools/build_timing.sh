#!/bin/bash
# Builds the diagnostic library (per-phase shader-clock stamps in the search kernel) into tools/_diag/.
set -e
cd "$(dirname "$0")/.."
B=vectordb_amd/_csrc_timing; rm -rf $B; mkdir -p $B tools/_diag
cp vectordb_amd/csrc/*.hip vectordb_amd/csrc/*.cpp vectordb_amd/csrc/*.hpp vectordb_amd/csrc/Makefile $B/
make -C $B -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DVDB_TIMING -Wno-unused-result -Wno-pass-failed" libvdbmi.so >/dev/null
cp $B/libvdbmi.so tools/_diag/libvdbmi_timing.so
rm -rf $B
echo built tools/_diag/libvdbmi_timing.so
