#!/bin/bash
# Dev tool: depthwise rows-per-thread sweep on the MobileNetV2 workload (roofline class = depthwise, live event timing)
for th in 0 2 4 8 16; do
  echo "dw_th=$th: $(PCV_BENCH_TUNE=dw_th=$th python bench.py --workload mobilenetv2_w1_bs512 --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['achieved'], r['avg_launch_us'], {k: (v['avg_us'], v['gbs']) for k,v in r['per_shape'].items()})")"
done
