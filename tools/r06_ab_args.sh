#!/bin/bash
# A/B of bench.py arguments on one lease, 3 interleaved repeats (bash tools/r06_ab_args.sh out "--threads 16" "--threads 32" ...)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; shift
mkdir -p $OUT
WL=/dev/shm/mpa_wl
python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > /dev/null 2>&1
for rep in 1 2 3; do
  k=0
  for v in "$@"; do
    k=$((k+1))
    python bench.py --no-extra --no-cpu-baseline --no-cli --steps 8 --warmup 2 --workload-cache $WL $v > $OUT/v${k}_$rep.json 2> $OUT/v${k}_$rep.err
    echo "[$v] rep $rep: $(python tools/bench_line.py $OUT/v${k}_$rep.json | head -3 | tr '\n' ' ' | cut -c1-330)"
  done
done
