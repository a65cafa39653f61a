"""Dev tool: merge gpurun_out/rNN_<workload>_<dtype>_pmc.json files (tests/tools/sh/round_profiles.sh) into profiles/pmc_traffic.json.
Entries of workloads that are not among the files passed are KEPT (the file is updated, not replaced).

    python tests/tools/merge_pmc.py <commit> gpurun_out/r04_*_pmc.json
"""
import json, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
commit = sys.argv[1]
cur = json.load(open(dst)) if os.path.exists(dst) else {}
for p in sys.argv[2:]:
    m = re.match(r"r(\d+)_(.+)_(bf16|fp16|fp32)_pmc\.json$", os.path.basename(p))
    rnd, w, dtype = int(m.group(1)), m.group(2), m.group(3)
    cur[w] = dict(dtype=dtype, classes=json.load(open(p)),
                  note="FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE exact; separate rocprofv3 --pmc passes over "
                       "`PCV_BENCH_PROFILE=1 python3 bench.py --workload {} --steps 3 --warmup 1 --no-cpu-baseline` (every dispatch a "
                       "full-batch launch), round {}, commit {}, one MI355X; tests/tools/sh/round_profiles.sh".format(w, rnd, commit))
json.dump(cur, open(dst, "w"), indent=1)
print("wrote", dst, sorted(cur))
