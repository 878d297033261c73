#!/bin/bash
# Round 3, GPU call 10: the whole refinement on the device -- parity, then the stream.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03j}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 1200 python -m pytest tests/test_map_gpu.py tests/test_seed_gpu.py -m gpu -q -x) > $OUT/pytest_map_seed.log 2>&1; grep -h "passed\|failed\|error\|Error" $OUT/pytest_map_seed.log | tail -5
(time timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest_scale.log 2>&1; grep -h "passed\|failed\|error\|Error" $OUT/pytest_scale.log | tail -3
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --no-extra --workload-cache $WL --steps 20 --warmup 5 $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "MPA_TIMING=1" ""
grep "mpa-cpu" $OUT/base.err | tail -9
python tools/timing_agg.py < $OUT/base.err > $OUT/timing_base.txt
grep -E "refine|plan:|A\.|stage A" $OUT/timing_base.txt
run hostref "MPA_GPU_REFINE_CHAINS=0" "--no-cpu-baseline"
run p1 "MPA_PLANNERS=1" "--no-cpu-baseline"
run p3 "MPA_PLANNERS=3" "--no-cpu-baseline"
run t12 "" "--no-cpu-baseline --threads 12"
rm -f $OUT/*.err
du -sh $OUT
