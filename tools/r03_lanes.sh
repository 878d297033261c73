#!/bin/bash
# Round 3, GPU call 7: the DP lanes bound the stream -- threads (quota throttling?), lanes, pool growth; hybrid chain forward tests.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03g}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_seed_gpu.py -m gpu -q -x) > $OUT/pytest_seed.log 2>&1; tail -4 $OUT/pytest_seed.log | head -2
(time timeout 1200 python -m pytest tests/test_map_gpu.py -m gpu -q -x) > $OUT/pytest_map.log 2>&1; tail -4 $OUT/pytest_map.log | head -2
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/base.err > $OUT/timing_base.txt
grep -E "dp:|mpa_dp_run|stage A1|chains on" $OUT/timing_base.txt
run t8 "" "--no-cpu-baseline --threads 8"
run t12 "" "--no-cpu-baseline --threads 12"
run t16 "" "--no-cpu-baseline --threads 16"
run t16l5 "MPA_DP_LANES=5" "--no-cpu-baseline --threads 16"
run t16l6 "MPA_DP_LANES=6" "--no-cpu-baseline --threads 16"
run t16l6s3 "MPA_DP_LANES=6 MPA_SEEDERS=3" "--no-cpu-baseline --threads 16"
run t12l6 "MPA_DP_LANES=6" "--no-cpu-baseline --threads 12"
run s40 "" "--no-cpu-baseline --steps 40 --warmup 10"
rm -f $OUT/*.err
du -sh $OUT
