#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03m}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 1200 python -m pytest tests/test_map_gpu.py tests/test_scale_gpu.py tests/test_dist_gpu.py -m gpu -q -x) > $OUT/pytest.log 2>&1; grep -h "passed\|failed\|error\|Error" $OUT/pytest.log | tail -3
run() {
  ( env $2 timeout 600 python bench.py --no-extra --workload-cache $WL --steps 20 --warmup 5 $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run tail4 "" ""
run nosplit "MPA_EDGE_SPLIT=1" "--no-cpu-baseline"
run tail4b "" "--no-cpu-baseline"
run nosplitb "MPA_EDGE_SPLIT=1" "--no-cpu-baseline"
run tail2 "MPA_EDGE_SPLIT=2" "--no-cpu-baseline"
run tail4c "" "--no-cpu-baseline"
run trace "MPA_TRACE=1" "--no-cpu-baseline"
python tools/gantt.py $OUT/trace.err > $OUT/gantt.txt 2>&1; head -16 $OUT/gantt.txt
rm -f $OUT/*.err
