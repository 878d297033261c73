#!/bin/bash
# Round 3, GPU call 16: the 32-bit operator on the device (k_gs32), the stream with the splat-int16 traceback and without the
# pool equalisation of call 15, and edge splitting once more now that a part sizes its pools for the whole batch.
#   bash tools/r03_gs32.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03p}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
PREV=$PWD/miniprot_amd/libmpamd_prev.so
(time timeout 600 python -m pytest tests/test_compat.py -m gpu -q -x -k gs32) > $OUT/pytest_gs32.log 2>&1; tail -4 $OUT/pytest_gs32.log
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
run() {   # label, env, bench args
  ( env $2 timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run new "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/new.err > $OUT/timing_new.txt
run prev "MPA_LIB_PATH=$PREV" "--no-cpu-baseline"
run edge4 "MPA_EDGE_SPLIT=4 MPA_EDGE_SPLIT_HEAD=1" "--no-cpu-baseline"
run new2 "" "--no-cpu-baseline"
run edge4t "MPA_EDGE_SPLIT=4" "--no-cpu-baseline"
run edge2 "MPA_EDGE_SPLIT=2 MPA_EDGE_SPLIT_HEAD=1" "--no-cpu-baseline"
rm -f $OUT/*.err
du -sh $OUT
