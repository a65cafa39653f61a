#!/bin/bash
# Dev tool (run HERE, on the build container): libraries with d3i_kernel built with extra flags under pytorchcv_amd/csrc/ab/ for same-box
# A/B and in-kernel stamps on the GPU box:  tests/tools/sh/d3i_variants.sh cyc "-DD3I_CYCLES" nord "-DD3I_DBG=1" ...   then on the box:
#   python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_d3i_cyc.so tests/tools/d3i_cycles.py
set -e
cd "$(dirname "$0")/../../../pytorchcv_amd/csrc"
mkdir -p ab
OBJS=$(ls *.o | grep -v '^d3i_')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c d3i_16bit.hip -o ab/d3i_16bit_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpcv_amd_d3i_$name.so $OBJS ab/d3i_16bit_$name.o
  ) &
done
wait
ls -la ab/*d3i*.so
