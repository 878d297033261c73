#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03l}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 900 python -m pytest tests/test_seed_gpu.py tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest.log 2>&1; grep -h "passed\|failed\|error\|Error" $OUT/pytest.log | tail -3
run() {
  ( env $2 timeout 600 python bench.py --no-extra --workload-cache $WL --steps 20 --warmup 5 $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run cap4096 "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/cap4096.err | grep -E "seed:|stage A1"
run cap2048 "MPA_TIMING=1 MPA_SIFT_CAP=2048" "--no-cpu-baseline"
python tools/timing_agg.py < $OUT/cap2048.err | grep -E "seed:|stage A1"
run cap4096b "" "--no-cpu-baseline"
run cap2048b "MPA_SIFT_CAP=2048" "--no-cpu-baseline"
run s40 "" "--no-cpu-baseline --steps 40 --warmup 10"
rm -f $OUT/*.err
