"""Summarise one forward from a rocprofv3 kernel_trace.csv: per-launch durations in dispatch order (dev tool)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a forward starts at the stem kernel (which reads the fp32 NCHW input itself since round 2) or at the layout kernel in front of it
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name'] or 'stem_conv_kernel' in r['Kernel_Name']]
idx = [i for k, i in enumerate(idx) if k == 0 or i != idx[k - 1] + 1]          # (layout kernel + stem = one start)
a, b = idx[-2], idx[-1]
tot = 0
for r in rows[a:b]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot += d
    print('%-62s grid=%-8s vgpr=%-4s %8.1f us' % (r['Kernel_Name'][:60], r['Grid_Size_X'], r['VGPR_Count'], d / 1e3))
print('sum %.1f us, span %.1f us' % (tot / 1e3, (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3))
