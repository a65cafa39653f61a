# LDS / MFMA counters per kernel over one workload (dev tool): bash tests/tools/sh/pmc_net.sh <workload>
set -e
R=$GRAFT_REPO_ROOT
W=$1
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcn_$i -o p -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --graph 0 > $R/gpurun_out/pmcn_$i.log 2>&1
  python3 $R/tests/tools/pmc_summary.py $R/gpurun_out/pmcn_$i/p_counter_collection.csv > $R/gpurun_out/pmcn_${W}_$i.txt
  rm -rf $R/gpurun_out/pmcn_$i
done
