"""Dev tool: merge gpurun_out/r03_<workload>_<dtype>_pmc.json files (tests/tools/sh/round_profiles.sh) into profiles/pmc_traffic.json.

    python tests/tools/merge_pmc.py <commit> gpurun_out/r03_*_pmc.json
"""
import json, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
commit = sys.argv[1]
cur = {}
for p in sys.argv[2:]:
    m = re.match(r"r03_(.+)_(bf16|fp16|fp32)_pmc\.json$", os.path.basename(p))
    w, dtype = m.group(1), m.group(2)
    cur[w] = dict(dtype=dtype, classes=json.load(open(p)),
                  note="FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE exact; separate rocprofv3 --pmc passes over "
                       "`PCV_BENCH_PROFILE=1 python3 bench.py --workload {} --steps 3 --warmup 1 --no-cpu-baseline` (every dispatch a "
                       "full-batch launch), round 3, commit {}, one MI355X; tests/tools/sh/round_profiles.sh".format(w, commit))
json.dump(cur, open(dst, "w"), indent=1)
print("wrote", dst, sorted(cur))
