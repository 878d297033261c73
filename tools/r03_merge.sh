#!/bin/bash
# Round 3, GPU call 21: the DP rounds of several planned batches in one launch per round (MPA_DP_MERGE).   bash tools/r03_merge.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03t}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests -m gpu -q -x) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed\|Error\|Abort" $OUT/pytest_gpu.log | tail -3
run() {
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]
    print("      launches", r["launches"], "avg ms", round(r["avg_launch_ms"],1), "frac/launch", round(r["frac"],4), "wall frac", round(r["wall"]["frac"],4))
except Exception as e: print("      no record", repr(e)[:100])
PY
  grep -h "rror" $OUT/$1.err | head -2
}
run base "" ""
run m2 "MPA_DP_MERGE=2" ""
run m2l3 "MPA_DP_MERGE=2 MPA_DP_LANES=3" ""
run m3l2 "MPA_DP_MERGE=3 MPA_DP_LANES=2" ""
run base2 "" ""
rm -f $OUT/*.err
