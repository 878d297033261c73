#!/bin/bash
# Round 3, GPU call 4: both chaining rounds on the device -- parity tests, then what it does to the host CPU and the stream.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03d}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_seed_gpu.py -m gpu -q -x) > $OUT/pytest_seed.log 2>&1; tail -4 $OUT/pytest_seed.log | head -2
(time timeout 1200 python -m pytest tests/test_map_gpu.py tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest_map_scale.log 2>&1; tail -4 $OUT/pytest_map_scale.log | head -2
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "MPA_TIMING=1" ""
grep "mpa-cpu" $OUT/base.err | tail -10
python tools/timing_agg.py < $OUT/base.err > $OUT/timing_base.txt
grep -E "seed:|A1|A\.|plan:" $OUT/timing_base.txt
run prof "MPA_EXTRACT_PROF=1" "--no-cpu-baseline --steps 6 --warmup 2"
grep "extract-prof" $OUT/prof.err | tail -6
run hostchains "MPA_DEVICE_CHAINS=0" "--no-cpu-baseline"
run seed3 "MPA_SEEDERS=3" "--no-cpu-baseline"
run seed4l6 "MPA_SEEDERS=4 MPA_DP_LANES=6" "--no-cpu-baseline"
run trace "MPA_TRACE=1 MPA_SEEDERS=3" "--no-cpu-baseline"
python tools/gantt.py $OUT/trace.err > $OUT/gantt.txt 2>&1; head -14 $OUT/gantt.txt
rm -rf $OUT/trace; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -24 $OUT/kernel_trace_stats.txt
rm -rf $OUT/trace $OUT/*.err
du -sh $OUT
