#!/bin/bash
# Dev tool: A/B of the depthwise kernels' tuning flags on one box (interleaved runs): dw_flags bit 0 = non-temporal stores, bit 1 = blocks in
# dispatch order (no XCD remap). Prints images/s, GB/s of the depthwise class and its average launch.
for rep in 1 2; do for t in ${DW_FLAGS:-"dw_flags=0" "dw_flags=2"}; do
  echo "$t: $(PCV_BENCH_TUNE=$t python bench.py --workload ${WORKLOAD:-mobilenetv2_w1_bs512} --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['achieved'], r['avg_launch_us'])")"
done; done
