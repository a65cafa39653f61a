#!/bin/bash
# Dev tool (GPU box): time the MBR_DBG variant libraries built by mbr_variants.sh (results of those builds are wrong by construction)
export BENCH_MBW_ONLY=default
U=${UNITS:-"32:16:112:1:0 24:24:56:1:1 32:32:28:1:1"}
echo "== base"; python tests/tools/bench_mbw.py fp16 512 $U 2>&1 | grep -v amdgpu.ids
for n in "$@"; do echo "== dbg $n"; python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_mbr$n.so tests/tools/bench_mbw.py fp16 512 $U 2>&1 | grep -v amdgpu.ids; done
