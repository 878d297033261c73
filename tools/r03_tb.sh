#!/bin/bash
# Round 3, GPU call 15: the splat-int16 traceback sweep (glob_narrow / glob_wide_body), the hand-off retry and the pool
# equalisation -- parity first, then ns/row and the stream against the previous build (miniprot_amd/libmpamd_prev.so, built
# from the previous commit; not tracked).
#   bash tools/r03_tb.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03o}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
PREV=$PWD/miniprot_amd/libmpamd_prev.so
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
# traceback ns/row: lone wave and saturated, previous build vs this one
for spec in "12 1" "12 4096" "28 1" "28 2048" "60 1" "60 2048" "120 1" "120 512" "250 1" "250 256" "1000 1" "1000 64"; do
  set -- $spec
  a=$(MPA_LIB_PATH=$PREV timeout 300 python tools/time_wide.py $1 20000 $2 1 2>&1 | tail -1)
  b=$(timeout 300 python tools/time_wide.py $1 20000 $2 1 2>&1 | tail -1)
  echo "prev: $a" | tee -a $OUT/traceback_ns_per_row.txt
  echo "new : $b" | tee -a $OUT/traceback_ns_per_row.txt
done
run() {   # label, env, bench args
  ( env $2 timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run new "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/new.err > $OUT/timing_new.txt
run prev "MPA_LIB_PATH=$PREV" "--no-cpu-baseline"
run new2 "" "--no-cpu-baseline"
run edge4 "MPA_EDGE_SPLIT=4 MPA_EDGE_SPLIT_HEAD=1" "--no-cpu-baseline"
run edge4t "MPA_EDGE_SPLIT=4" "--no-cpu-baseline"
run prev2 "MPA_LIB_PATH=$PREV" "--no-cpu-baseline"
rm -f $OUT/*.err
du -sh $OUT
