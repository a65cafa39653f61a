"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (dev tool): mean of each counter per dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    key = name[:40] + " grid=" + r.get("Grid_Size", "?")
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    print("   ", "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())), " n=%d" % len(next(iter(cs.values()))))
