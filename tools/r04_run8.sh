#!/bin/bash
# round 4, GPU call 8: config 5 (long introns, -G 500k): traceback budget per launch x mini-batch size
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
W=/dev/shm/mpa_wl
for spec in "2000 8192" "2000 16384" "2000 32768" "2000 65536" "1000 32768" "500 32768"; do
  set -- $spec
  MPA_TB_BUDGET_MB=$2 python bench.py --config 5 --steps 3 --warmup 1 --batch $1 --no-cpu-baseline --no-cli --workload-cache $W > $O/c5_b$1_tb$2.json 2> $O/c5_b$1_tb$2.err
  python tools/bench_line.py $O/c5_b$1_tb$2.json "config5 batch $1 tb $2 MB"
done
MPA_TIMING=1 MPA_TB_BUDGET_MB=32768 python bench.py --config 5 --steps 2 --warmup 1 --batch 2000 --no-cpu-baseline --no-cli --workload-cache $W > /dev/null 2> $O/timing.err
python tools/timing_agg.py < $O/timing.err | grep -E "dp:|mpa_dp_run|take|stage A|seeding on|refinement" > $O/c5_stage_clocks.txt; cat $O/c5_stage_clocks.txt
rm -f $O/*.err
rm -rf $W
