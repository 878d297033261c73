#!/bin/bash
# Round 3, GPU call 18: device pools that retire their old block inside a stream (freed when the stream is over) instead of
# hipFree()ing it on the spot -- does a stream still grow pools after the warm-up, and what does it cost?   bash tools/r03_retire.sh [outdir]
# (the experiment's outcome: retiring is slower, profiles/r03_experiments.txt; the MPA_POOL_RETIRE switch existed only in the build this ran on)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03r}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_map_gpu.py tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest_map.log 2>&1; grep -h "passed\|failed" $OUT/pytest_map.log | tail -2
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run r1t "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/r1t.err | grep -i "pool growth\|dp: buffers\|mpa_dp_run\|stage A \|stage A1" 
run r0t "MPA_TIMING=1 MPA_POOL_RETIRE=0" ""
python tools/timing_agg.py < $OUT/r0t.err | grep -i "pool growth\|dp: buffers\|mpa_dp_run\|stage A \|stage A1"
run r1a "" ""
run r0a "MPA_POOL_RETIRE=0" ""
run r1b "" ""
run r0b "MPA_POOL_RETIRE=0" ""
run r1c "" ""
run r0c "MPA_POOL_RETIRE=0" ""
rm -f $OUT/*.err
du -sh $OUT
