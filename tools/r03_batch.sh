#!/bin/bash
# Round 3, GPU call 8: grouped placement in the sort replay (tests + profile), then mini-batch size x stage counts.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03h}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_seed_gpu.py tests/test_map_gpu.py -m gpu -q -x) > $OUT/pytest_seed_map.log 2>&1; grep -h "passed\|failed\|error" $OUT/pytest_seed_map.log | tail -2
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "MPA_DP_LANES=5" "--steps 20 --warmup 5"
run prof "MPA_EXTRACT_PROF=1" "--no-cpu-baseline --steps 6 --warmup 2"
grep "extract-prof" $OUT/prof.err | tail -4
run b5000 "MPA_DP_LANES=5" "--no-cpu-baseline --batch 5000 --steps 16 --warmup 4"
run b6667 "MPA_DP_LANES=5" "--no-cpu-baseline --batch 6667 --steps 12 --warmup 3"
run b10000 "MPA_DP_LANES=5" "--no-cpu-baseline --batch 10000 --steps 8 --warmup 2"
run b10000s3 "MPA_DP_LANES=5 MPA_SEEDERS=3 MPA_PLANNERS=3" "--no-cpu-baseline --batch 10000 --steps 8 --warmup 2"
run b6667s3 "MPA_DP_LANES=5 MPA_SEEDERS=3 MPA_PLANNERS=3" "--no-cpu-baseline --batch 6667 --steps 12 --warmup 3"
run b4000s3p3 "MPA_DP_LANES=5 MPA_SEEDERS=3 MPA_PLANNERS=3" "--no-cpu-baseline --steps 20 --warmup 5"
rm -f $OUT/*.err
du -sh $OUT
