# PMC passes over the p1r kernel on one or two layers (dev tool): bash tests/tools/sh/pmc_p1r.sh [cfg]
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-$R/tests/tools/bench_p1r_exp.cfg}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_$i -o p -- python3 $R/tests/tools/bench_conv.py $CFG > $R/gpurun_out/pmcp_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmcp_$i.log; continue; }
  F=$(ls $R/gpurun_out/pmcp_$i/*/p_counter_collection.csv $R/gpurun_out/pmcp_$i/p_counter_collection.csv 2>/dev/null | head -1)
  python3 $R/tests/tools/pmc_summary.py $F p1r_kernel
  rm -rf $R/gpurun_out/pmcp_$i
done
