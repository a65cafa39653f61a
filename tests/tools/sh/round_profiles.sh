# Per-round profile artifacts (RND=r04 by default) (on the GPU box, from the repo root; `bash tests/tools/sh/round_profiles.sh [workload:dtype ...]`):
#   gpurun_out/${RND}_<workload>_<dtype>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
#   gpurun_out/${RND}_<workload>_<dtype>_pmc.json           per-class HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes)
#   gpurun_out/${RND}_<workload>_<dtype>_pmc_{fetch,write}_by_kernel.txt
# bench.py runs with PCV_BENCH_PROFILE=1: full-batch eager forwards only, one lane, so that every dispatch in the profiler's tables
# is a full-batch launch (bench.py reads the all-launch class averages from the stats file). Copy into profiles/ what is to be judged;
# merge the pmc.json files into profiles/pmc_traffic.json with tests/tools/merge_pmc.py.
set -e
RND=${RND:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCV_BENCH_PROFILE=1
SPECS=${@:-resnet50_bs256:bf16 mobilenetv2_w1_bs512:fp16 resnext101_32x4d_bs256:bf16}
for spec in $SPECS; do
  W=${spec%%:*}; D=${spec##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o p -- python3 $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$W.log 2>&1
  cp $R/gpurun_out/prof_$W/*/p_kernel_stats.csv $R/gpurun_out/${RND}_${W}_${D}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/prof_$W/p_kernel_stats.csv $R/gpurun_out/${RND}_${W}_${D}_kernel_stats.csv
  echo "$W stats done"; tail -1 $R/gpurun_out/prof_$W.log | cut -c1-200
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${W}_$C -o p -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${W}_$C.log 2>&1
    echo "$W $C done"
  done
  F=$(ls $R/gpurun_out/pmc_${W}_FETCH_SIZE/*/p_counter_collection.csv $R/gpurun_out/pmc_${W}_FETCH_SIZE/p_counter_collection.csv 2>/dev/null | head -1)
  G=$(ls $R/gpurun_out/pmc_${W}_WRITE_SIZE/*/p_counter_collection.csv $R/gpurun_out/pmc_${W}_WRITE_SIZE/p_counter_collection.csv 2>/dev/null | head -1)
  python3 $R/tests/tools/pmc_class_traffic.py $F $G > $R/gpurun_out/${RND}_${W}_${D}_pmc.json
  cat $R/gpurun_out/${RND}_${W}_${D}_pmc.json
  python3 $R/tests/tools/pmc_summary.py $F > $R/gpurun_out/${RND}_${W}_${D}_pmc_fetch_by_kernel.txt 2>/dev/null || true
  python3 $R/tests/tools/pmc_summary.py $G > $R/gpurun_out/${RND}_${W}_${D}_pmc_write_by_kernel.txt 2>/dev/null || true
  # keep the merged-back files small: the per-dispatch tables are tens of MB
  rm -rf $R/gpurun_out/pmc_${W}_FETCH_SIZE $R/gpurun_out/pmc_${W}_WRITE_SIZE
  rm -f $R/gpurun_out/prof_$W/*/p_kernel_trace.csv $R/gpurun_out/prof_$W/p_kernel_trace.csv
done
