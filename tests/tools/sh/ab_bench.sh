# Same-box A/B of two library builds (on the GPU box, from the repo root): `bash tests/tools/sh/ab_bench.sh <old.so> workload...`
OLD=$1; shift
for W in "$@"; do
  for rep in 1 2; do
    python bench.py --workload $W --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W new', d['value'], d['ms_per_step'])"
    python tests/tools/ab_lib.py $OLD bench.py --workload $W --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W old', d['value'], d['ms_per_step'])"
  done
done
