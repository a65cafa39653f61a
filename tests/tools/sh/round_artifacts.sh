# Per-round artifacts (RND=r04 by default) in ONE call (on the GPU box, from the repo root; `PCV_COMMIT=<hash> bash tests/tools/sh/round_artifacts.sh`):
#   1. round_profiles.sh: rocprofv3 kernel stats + the two PMC passes of the three headline workloads (PCV_BENCH_PROFILE=1);
#   2. those files copied into this snapshot's profiles/ and merged into profiles/pmc_traffic.json, so that
#   3. the bench.py lines of all nine workloads (gpurun_out/${RND}_bench_<workload>.json) carry rocprof / traffic figures of the SAME
#      binary. Back in the container: cp gpurun_out/${RND}_* profiles/ ; cp gpurun_out/pmc_traffic.json profiles/ ; commit.
set -e
export RND=${RND:-r04}
R=$GRAFT_REPO_ROOT
bash $R/tests/tools/sh/round_profiles.sh
cd $R
cp gpurun_out/${RND}_*_kernel_stats.csv gpurun_out/${RND}_*_by_kernel.txt profiles/
python3 tests/tools/merge_pmc.py ${PCV_COMMIT:-unrecorded} gpurun_out/${RND}_*_pmc.json
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python3 bench.py > gpurun_out/${RND}_bench_resnet50_bs256.json 2> gpurun_out/bench_resnet50_bs256.err
cut -c1-230 gpurun_out/${RND}_bench_resnet50_bs256.json
for W in mobilenetv2_w1_bs512 resnext101_32x4d_bs256 resnet18_bs256 seresnet50_bs256 mobilenetv3_large_w1_bs512 efficientnet_b0_bs256 vgg16_bs128 seresnext50_32x4d_bs256; do
  python3 bench.py --workload $W --no-cpu-baseline > gpurun_out/${RND}_bench_$W.json 2> gpurun_out/bench_$W.err
  cut -c1-230 gpurun_out/${RND}_bench_$W.json
done
