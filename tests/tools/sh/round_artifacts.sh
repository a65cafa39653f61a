# Round artifacts (on the GPU box, from the repo root): rocprofv3 kernel stats of the three headline workloads (single lane) and
# the bench.py JSON lines. Outputs under gpurun_out/; copy what is to be judged into profiles/.
set -e
R=$GRAFT_REPO_ROOT
bash $R/tests/tools/sh/prof_workloads.sh
cd $R
for W in resnet50_bs256 mobilenetv2_w1_bs512 resnext101_32x4d_bs256; do
  python3 tests/tools/trace_summary.py gpurun_out/prof_$W/p_kernel_trace.csv | cut -c1-120 > gpurun_out/per_launch_$W.txt || true
done
python3 bench.py > gpurun_out/bench_resnet50_bs256.json 2> gpurun_out/bench_resnet50_bs256.err
cut -c1-230 gpurun_out/bench_resnet50_bs256.json
for W in mobilenetv2_w1_bs512 resnext101_32x4d_bs256 resnet18_bs256 seresnet50_bs256 mobilenetv3_large_w1_bs512 efficientnet_b0_bs256 vgg16_bs128 seresnext50_32x4d_bs256; do
  python3 bench.py --workload $W --no-cpu-baseline > gpurun_out/bench_$W.json 2> gpurun_out/bench_$W.err
  cut -c1-230 gpurun_out/bench_$W.json
done
