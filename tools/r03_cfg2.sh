#!/bin/bash
# Round 3, GPU call 20: config 2 (50 Mbp x 1 000 proteins) with the build before the second half of the round and with this one.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03s}
mkdir -p $OUT
PREV=$PWD/miniprot_amd/libmpamd_prev.so
run() {
  ( env $2 timeout 300 python bench.py --config 2 --steps 40 --warmup 10 --no-cpu-baseline --workload-cache /dev/shm/mpa_wl2 $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run new1 "" ""
run prev1 "MPA_LIB_PATH=$PREV" ""
run new2 "" ""
run prev2 "MPA_LIB_PATH=$PREV" ""
run new3 "MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/new3.err | grep -i "pool growth\|dp: \|mpa_dp_run\|stage A " | head -20
run prev3 "MPA_LIB_PATH=$PREV MPA_TIMING=1" ""
python tools/timing_agg.py < $OUT/prev3.err | grep -i "dp: \|mpa_dp_run\|stage A " | head -20
run new4 "" "--steps 120 --warmup 20"
run prev4 "MPA_LIB_PATH=$PREV" "--steps 120 --warmup 20"
rm -f $OUT/*.err
