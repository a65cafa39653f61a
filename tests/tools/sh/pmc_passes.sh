# HBM traffic per launch of each workload's roofline kernel class: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# eager bench runs, summarised by tests/tools/pmc_class_traffic.py into gpurun_out/pmc_<workload>.json (merge into profiles/pmc_traffic.json).
#   bash tests/tools/sh/pmc_passes.sh [workload:class:skip ...]            (on the GPU box, from the repo root)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# workload:class:launches of the class in one forward (the first forward of bench.py runs 8 images: skipped)
SPECS=${@:-resnet50_bs256:dense3x3:16 mobilenetv2_w1_bs512:fused_unit:10 resnext101_32x4d_bs256:grouped3x3:33}
for spec in $SPECS; do
  W=${spec%%:*}; rest=${spec#*:}; K=${rest%%:*}; SKIP=${rest##*:}
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${W}_$C -o p -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --graph 0 > $R/gpurun_out/pmc_${W}_$C.log 2>&1
    echo "$W $C done"
  done
  python3 $R/tests/tools/pmc_class_traffic.py $R/gpurun_out/pmc_${W}_FETCH_SIZE/p_counter_collection.csv $R/gpurun_out/pmc_${W}_WRITE_SIZE/p_counter_collection.csv $K $SKIP > $R/gpurun_out/pmc_$W.json
  cat $R/gpurun_out/pmc_$W.json
  # keep the merged-back files small: the per-dispatch counter tables are tens of MB
  python3 $R/tests/tools/pmc_summary.py $R/gpurun_out/pmc_${W}_FETCH_SIZE/p_counter_collection.csv > $R/gpurun_out/pmc_${W}_fetch_by_kernel.txt 2>/dev/null || true
  python3 $R/tests/tools/pmc_summary.py $R/gpurun_out/pmc_${W}_WRITE_SIZE/p_counter_collection.csv > $R/gpurun_out/pmc_${W}_write_by_kernel.txt 2>/dev/null || true
  rm -rf $R/gpurun_out/pmc_${W}_FETCH_SIZE $R/gpurun_out/pmc_${W}_WRITE_SIZE
done
