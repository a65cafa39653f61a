set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in resnet50_bs256 mobilenetv2_w1_bs512 resnext101_32x4d_bs256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o p -- python3 $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 > $R/gpurun_out/prof_$W.log 2>&1
  echo "$W done"; tail -1 $R/gpurun_out/prof_$W.log | cut -c1-300
done
