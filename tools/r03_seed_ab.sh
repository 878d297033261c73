#!/bin/bash
# Round 3, GPU call 1: parity of the new seeding path + A/B of the three seeding modes at config 3 (one box, one cached workload).
#   bash tools/r03_seed_ab.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03a}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
# 1. GPU tests (seeding first: the new kernel)
(time timeout 900 python -m pytest tests/test_seed_gpu.py -m gpu -q -x) > $OUT/pytest_seed.log 2>&1; tail -3 $OUT/pytest_seed.log
(time timeout 1200 python -m pytest tests/test_scale_gpu.py tests/test_map_gpu.py -m gpu -q) > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
# 2. config 3: sort (with the reference run: identity), census, sift
for mode in ${MODES:-sift}; do
  extra="--no-cpu-baseline"; [ $mode = ${REFMODE:-sift} ] && extra=""
  ( MPA_SEED_MODE=$mode MPA_TIMING=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --workload-cache $WL $extra ) > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  python tools/timing_agg.py < $OUT/bench_$mode.err > $OUT/timing_$mode.txt; rm -f $OUT/bench_$mode.err
  python tools/bench_line.py $mode $OUT/bench_$mode.json
  grep -E "seed:|seeding on the GPU|A1" $OUT/timing_$mode.txt
done
# 2b. with cheap seeding, what do the pipeline knobs do?  (same cached workload, md5 must stay)
k=0
for spec in "MPA_TRACE=1" "MPA_DP_LANES=6" "MPA_SEEDERS=1" "MPA_PLANNERS=3" "MPA_DP_LANES=6 MPA_PLANNERS=3" "MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_ARENA_MAX=64"; do
  ( env $spec timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --workload-cache $WL ) > $OUT/knob$k.json 2> $OUT/knob$k.err
  python tools/bench_line.py "$spec" $OUT/knob$k.json
  [ $k = 0 ] && python tools/gantt.py $OUT/knob0.err > $OUT/gantt_sift.txt 2>&1
  rm -f $OUT/knob$k.err
  k=$((k+1))
done
head -12 $OUT/gantt_sift.txt
# 3. kernel trace of the default (sift) run
rm -rf $OUT/trace; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -30 $OUT/kernel_trace_stats.txt
rm -rf $OUT/trace
du -sh $OUT
