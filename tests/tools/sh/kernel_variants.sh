#!/bin/bash
# Dev tool (run HERE, on the build container): libraries with ONE translation unit of the library rebuilt with extra flags, under
# pytorchcv_amd/csrc/ab/, for same-box A/B and in-kernel stamps on the GPU box:
#   tests/tools/sh/kernel_variants.sh d1i_16bit cyc "-DD1I_CYCLES" nord "-DD1I_DBG=1" ...
# then on the box:  python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_d1i_16bit_cyc.so tests/tools/d1i_cycles.py
set -e
cd "$(dirname "$0")/../../../pytorchcv_amd/csrc"
mkdir -p ab
stem=$1; shift
OBJS=$(ls *.o | grep -v "^${stem}\.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c ${stem}.hip -o ab/${stem}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpcv_amd_${stem}_$name.so $OBJS ab/${stem}_$name.o
  ) &
done
wait
ls -la ab/*${stem}*.so
