#!/bin/bash
# Dev tool: A/B of tuning switches on one box (interleaved runs): MobileNetV2 / SE-ResNet-50 / PreResNet-like workloads
for rep in 1 2; do for t in "dw_flags=0" "dw_flags=1"; do
  echo "$t: $(PCV_BENCH_TUNE=$t python bench.py --workload mobilenetv2_w1_bs512 --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['achieved'], r['avg_launch_us'])")"
done; done
