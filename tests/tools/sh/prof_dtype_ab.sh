set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCV_BENCH_PROFILE=1
for D in fp16 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mv3_$D -o p -- python3 $R/bench.py --workload mobilenetv3_large_w1_bs512 --dtype $D --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_mv3_$D.log 2>&1
  cp $R/gpurun_out/prof_mv3_$D/*/p_kernel_stats.csv $R/gpurun_out/mv3_${D}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/prof_mv3_$D/p_kernel_stats.csv $R/gpurun_out/mv3_${D}_kernel_stats.csv
  rm -rf $R/gpurun_out/prof_mv3_$D
done
