#!/bin/bash
# round 4, GPU call 4: GPU tests (the stream pipeline has a sketch stage now), the bench, stage-count variants, stage clocks + Gantt
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
grep -h "passed\|failed" $O/pytest_gpu.log | tail -2
W=/dev/shm/mpa_wl
python bench.py --steps 10 --warmup 2 --workload-cache $W > $O/bench.json 2> $O/bench.err
python tools/bench_line.py $O/bench.json
run() {   # label, env, bench args
  ( env $2 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --no-cli --workload-cache $W $3 ) > $O/$1.json 2> $O/$1.err
  python tools/bench_line.py $O/$1.json "$1 [$2 $3]"
}
run s3 "MPA_SEEDERS=3" ""
run l6 "MPA_DP_LANES=6" ""
run s3l6 "MPA_SEEDERS=3 MPA_DP_LANES=6" ""
run s3l6p4 "MPA_SEEDERS=3 MPA_DP_LANES=6 MPA_PLANNERS=4" ""
run b5000 "" "--batch 5000"
( MPA_TRACE=1 MPA_TIMING=1 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-cli --workload-cache $W ) > $O/trace.json 2> $O/trace.err
python tools/gantt.py $O/trace.err > $O/pipeline_gantt.txt 2>&1; cat $O/pipeline_gantt.txt
{ python tools/timing_agg.py < $O/trace.err; grep "mpa-cpu" $O/trace.err | tail -14; } > $O/stage_clocks.txt 2>&1
tail -12 $O/stage_clocks.txt
rm -f $O/trace.err
rm -rf $W
