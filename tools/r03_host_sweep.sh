#!/bin/bash
# Round 3, GPU call 3: the box gives the process a 16-CPU quota -- what do blocking waits and the thread count do, and where does
# the host CPU time go (MPA_TIMING: [mpa-cpu] lines)?  One box, one cached workload.
#   bash tools/r03_host_sweep.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03c}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
bash tools/hostinfo.sh 2>&1 | head -3
(time timeout 900 python -m pytest tests/test_seed_gpu.py tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest_seed_scale.log 2>&1; tail -3 $OUT/pytest_seed_scale.log
run() {   # label, env, bench args
  ( env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $3 ) > $OUT/$1.json 2> $OUT/$1.err
  python tools/bench_line.py "$1 [$2 $3]" $OUT/$1.json
}
run base "MPA_TIMING=1" ""                                   # default threads (from the quota), blocking waits, with the reference run
grep "mpa-cpu" $OUT/base.err | tail -14
python tools/timing_agg.py < $OUT/base.err > $OUT/timing_base.txt
run spin "MPA_SPIN_WAIT=1" "--no-cpu-baseline"
run t12 "" "--no-cpu-baseline --threads 12"
run t16 "" "--no-cpu-baseline --threads 16"
run t32 "" "--no-cpu-baseline --threads 32"
run t64 "" "--no-cpu-baseline --threads 64"
run t16p1 "MPA_PLANNERS=1" "--no-cpu-baseline --threads 16"
run t24p3 "MPA_PLANNERS=3" "--no-cpu-baseline --threads 24"
run b2000 "" "--no-cpu-baseline --batch 2000 --steps 40 --warmup 10"
rm -f $OUT/*.err
du -sh $OUT
