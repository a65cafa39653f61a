# usage: bash tests/tools/sh/prof_one.sh <workload> [extra bench args]: rocprofv3 kernel trace + stats of a single-lane bench run
set -e
W=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o p -- python3 $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 "$@" > $R/gpurun_out/prof_$W.log 2>&1
grep -o '"value": [0-9.]*' $R/gpurun_out/prof_$W.log
cut -c1-100 $R/gpurun_out/prof_$W/p_kernel_stats.csv | head -14
