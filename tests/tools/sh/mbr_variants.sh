#!/bin/bash
# Dev tool (run HERE, on the build container): one library per MBR_DBG value under pytorchcv_amd/csrc/ab/ for subtract-a-component
# timing of mbr_kernel on the GPU box:  tests/tools/sh/mbr_variants.sh 1 2 4 ...   then on the box, per value n:
#   python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_mbr$n.so tests/tools/bench_mbw.py fp16 512 24:24:56:1:1
set -e
cd "$(dirname "$0")/../../../pytorchcv_amd/csrc"
mkdir -p ab
OBJS=$(ls *.o | grep -v '^mbr_')
for n in "$@"; do
  (
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DMBR_DBG=$n ${MBR_EXTRA} -c mbr_f16.hip -o ab/mbr_f16_$n.o &&
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DMBR_DBG=$n ${MBR_EXTRA} -c mbr_bf16.hip -o ab/mbr_bf16_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpcv_amd_mbr$n.so $OBJS ab/mbr_f16_$n.o ab/mbr_bf16_$n.o
  ) &
done
wait
ls -la ab/*.so
