"""Dev tool: per-launch HBM traffic of EVERY kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tests/tools/pmc_class_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>

The passes run bench.py with PCV_BENCH_PROFILE=1 (nothing but full-batch eager forwards), so every dispatch of a class counts.
Kernel names are mapped to classes by bench.kernel_class_of (the same mapping the bench line's `roofline.classes` uses).
Counters are reported in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes for 16 B/lane streaming
reads - MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16 B/lane stores. Prints one JSON object
{class: {launches: [n_fetch, n_write], fetch_mb_per_launch, write_mb_per_launch, traffic_mb_per_launch}}.
"""
import csv, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import kernel_class_of  # noqa: E402


def per_class(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = kernel_class_of(r["Kernel_Name"])
        if k is None:
            continue
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch, write = per_class(sys.argv[1], "FETCH_SIZE"), per_class(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    if not nf or not nw:
        continue
    fm, wm = 2 * (f / nf) * 1024 / 1e6, (w / nw) * 1024 / 1e6
    out[k] = dict(launches=[nf, nw], fetch_mb_per_launch=round(fm, 2), write_mb_per_launch=round(wm, 2),
                  traffic_mb_per_launch=round(fm + wm, 2))
print(json.dumps(out))
