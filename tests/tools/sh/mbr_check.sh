#!/bin/bash
# Dev tool (GPU box): parity tests of the fused units, the unit benchmark and the in-kernel stamps of mbr_kernel in one call.
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu -k "mbconv_fused and reg" > gpurun_out/t_mbr.log 2>&1; tail -3 gpurun_out/t_mbr.log
BENCH_MBW_ONLY=${BENCH_MBW_ONLY:-default,mbw} timeout -k 10 200 python tests/tools/bench_mbw.py fp16 512 ${UNITS:-32:16:112:1:0 24:24:56:1:1 32:32:28:1:1 32:64:14:1:1} > gpurun_out/b_mbr.log 2>&1; grep -v amdgpu.ids gpurun_out/b_mbr.log
if [ -f pytorchcv_amd/csrc/ab/libpcv_amd_mbr0.so ]; then timeout -k 10 300 python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_mbr0.so tests/tools/mbr_cycles.py > gpurun_out/cyc_mbr.log 2>&1; grep -v amdgpu.ids gpurun_out/cyc_mbr.log; fi
