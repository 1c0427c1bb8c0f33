#!/bin/bash
# Compiler's resource usage (registers, LDS, scratch, occupancy) of the benchmark's instantiation of the search kernel,
# search_kernel<0, 8, 4, 2, false, 2, 3>, compiled alone from the library's own headers (no GPU needed).
# usage: tools/kernel_resource_usage.sh <tag>
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r02}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed \
  -Ivectordb_amd/csrc -Iinclude -Rpass-analysis=kernel-resource-usage -c tools/search_kernel_one.hip -o /tmp/search_kernel_one.o \
  2>&1 | grep -E "remark" | sed 's/^.*remark: //' > profiles/${TAG}_search_kernel_resource_usage.txt
cat profiles/${TAG}_search_kernel_resource_usage.txt
