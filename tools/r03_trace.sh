#!/bin/bash
# kernel trace of the bench command (default settings) + the scale tests + the 2-rank HIP test
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03k}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_dist_gpu.py -m gpu -q) > $OUT/pytest_scale_dist.log 2>&1; grep -h "passed\|failed\|error\|Error" $OUT/pytest_scale_dist.log | tail -5
( timeout 600 python bench.py --no-extra --workload-cache $WL --steps 20 --warmup 5 --no-cpu-baseline ) > $OUT/base.json 2> /dev/null
python tools/bench_line.py base $OUT/base.json
rm -rf $OUT/trace; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -36 $OUT/kernel_trace_stats.txt | cut -c1-150
rm -rf $OUT/trace $OUT/*.err
