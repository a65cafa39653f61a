"""Dev tool: per-launch HBM traffic of one kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tests/tools/pmc_class_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <class> [skip]

class: dense3x3 (d3q_kernel + igemm_conv_kernel<..., 9>), fused_unit (mbw_kernel / mbconv_kernel), depthwise (dwconv_kernel), grouped3x3 (gconv3x3_kernel: stride 1 with 4..16 channels
per group, gconv3x3r_kernel: stride 2 / 32 channels per group - all 33 grouped launches of ResNeXt-101).
Counters are reported in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes for 16 B/lane streaming
reads - MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16 B/lane stores. The first `skip` matching launches
(warm-up / packing forwards) are dropped. Prints one JSON object.
"""
import csv, json, sys


def pick(name, klass):
    if klass == "grouped3x3":
        return "gconv3x3_kernel" in name or "gconv3x3r_kernel" in name
    if klass == "dense3x3":
        return ("d3q_kernel" in name and ", true>(D3Params)" not in name) or ("igemm_conv_kernel" in name and name.rstrip().rstrip(")").split("(")[0].rstrip().endswith(", 9>"))
    if klass == "fused_unit":
        return "mbw_kernel" in name or "mbconv_kernel" in name
    if klass == "depthwise":
        return "dwconv_kernel" in name
    raise SystemExit("unknown class " + klass)


def mean_kb(path, counter, klass, skip):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and pick(r["Kernel_Name"], klass)]
    vals = vals[skip:]
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_kb(sys.argv[1], "FETCH_SIZE", sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
write, nw = mean_kb(sys.argv[2], "WRITE_SIZE", sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
print(json.dumps(dict(kernel_class=sys.argv[3], launches=[nf, nw], fetch_mb_per_launch=round(2 * fetch * 1024 / 1e6, 2),
                      write_mb_per_launch=round(write * 1024 / 1e6, 2),
                      traffic_mb_per_launch=round((2 * fetch + write) * 1024 / 1e6, 2),
                      note="FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE exact; separate --pmc passes")))
