#!/bin/bash
# Round 3, GPU call 17: the seeding stage is the busiest (87 % with two seeders) -- more seeders / planners / lanes, with the
# build that also has the kept fill registers in ext_narrow's scan.   bash tools/r03_stages.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03q}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2
for spec in "12 1" "12 4096" "60 1" "60 2048" "250 1" "250 256"; do
  set -- $spec
  echo "prev: $(MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_prev.so timeout 300 python tools/time_wide.py $1 20000 $2 4 2>&1 | tail -1)" | tee -a $OUT/ext_ns_per_row.txt
  echo "new : $(timeout 300 python tools/time_wide.py $1 20000 $2 4 2>&1 | tail -1)" | tee -a $OUT/ext_ns_per_row.txt
done
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
  grep -h "error\|Error" $OUT/$1.err | head -2
}
run base "" ""
run s3 "MPA_SEEDERS=3" ""
run s3p4 "MPA_SEEDERS=3 MPA_PLANNERS=4" ""
run s3l6 "MPA_SEEDERS=3 MPA_DP_LANES=6" ""
run s4p4l6 "MPA_SEEDERS=4 MPA_PLANNERS=4 MPA_DP_LANES=6" ""
run base2 "" ""
run s3b "MPA_SEEDERS=3" ""
run s3_40 "MPA_SEEDERS=3" "--steps 40 --warmup 10"
run base_40 "" "--steps 40 --warmup 10"
rm -f $OUT/*.err
du -sh $OUT
