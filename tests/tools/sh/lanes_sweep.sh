# bench.py at 2 / 3 / 4 graph lanes on the three headline workloads (on the GPU box, from the repo root)
set -e
for W in resnet50_bs256 resnext101_32x4d_bs256 mobilenetv2_w1_bs512; do
  for L in 2 3 4; do
    python bench.py --workload $W --lanes $L --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W lanes $L', d['value'], d['ms_per_step'])"
  done
done
