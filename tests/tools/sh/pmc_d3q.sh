# PMC passes over one dense 3x3 layer (dev tool): bash tests/tools/sh/pmc_d3q.sh [cfg]
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-$R/tests/tools/bench_one.cfg}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcd_$i -o p -- python3 $R/tests/tools/bench_conv.py $CFG > $R/gpurun_out/pmcd_$i.log 2>&1
  python3 $R/tests/tools/pmc_summary.py $R/gpurun_out/pmcd_$i/p_counter_collection.csv conv_kernel
  python3 $R/tests/tools/pmc_summary.py $R/gpurun_out/pmcd_$i/p_counter_collection.csv d3q_kernel
  rm -rf $R/gpurun_out/pmcd_$i
done
