# rocprofv3 average duration of the kernels whose name contains $1 in the eager profile run of workload $2 (dev tool):
#   bash tests/tools/sh/prof_kernel.sh head_gemm resnet50_bs256 [PCV_BENCH_TUNE value]
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCV_BENCH_PROFILE=1
[ -n "$3" ] && export PCV_BENCH_TUNE=$3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pk -o p -- python3 $R/bench.py --workload $2 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
F=$(ls $R/gpurun_out/pk/*/p_kernel_stats.csv $R/gpurun_out/pk/p_kernel_stats.csv 2>/dev/null | head -1)
grep "$1" $F | cut -d, -f1-4
rm -rf $R/gpurun_out/pk
