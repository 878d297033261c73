#!/bin/bash
# Round 3, GPU call 9: whole GPU suite with the round's defaults, then resource knobs of the round kernel next to the new seeding kernels.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03i}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 1500 python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed\|error" $OUT/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-120
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --no-extra --workload-cache $WL --steps 20 --warmup 5 $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "" ""
run wg3 "MPA_ROUND_WG_PER_CU=3" "--no-cpu-baseline"
run wg5 "MPA_ROUND_WG_PER_CU=5" "--no-cpu-baseline"
run noprio "MPA_UNIT_PRIO=0" "--no-cpu-baseline"
run seg24k "MPA_SIFT_SEG=24576" "--no-cpu-baseline"
run seg96k "MPA_SIFT_SEG=98304" "--no-cpu-baseline"
run serial96 "MPA_CHAIN_SERIAL_RUN=96" "--no-cpu-baseline"
run p3 "MPA_PLANNERS=3" "--no-cpu-baseline"
rm -f $OUT/*.err
du -sh $OUT
