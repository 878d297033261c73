#!/bin/bash
# A/B of pipeline knobs on one lease: each variant 3 times, interleaved (bash tools/r06_ab.sh out "VAR=val,VAR2=val" ...)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; shift
mkdir -p $OUT
WL=/dev/shm/mpa_wl
python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > /dev/null 2>&1   # builds the workload cache
for rep in 1 2 3; do
  for v in "$@"; do
    tag=$(echo "$v" | tr ',= /.' '_____' | tail -c 60)
    env $(echo $v | tr ',' ' ') python bench.py --no-extra --no-cpu-baseline --no-cli --steps 8 --warmup 2 --workload-cache $WL > $OUT/${tag}_$rep.json 2> $OUT/${tag}_$rep.err
    echo "$v rep $rep: $(python tools/bench_line.py $OUT/${tag}_$rep.json | head -1 | cut -d' ' -f2-9)"
  done
done
