#!/bin/bash
# stream evidence: kernel trace of the bench command (in-stream and nearly alone), pipeline Gantt, stage clocks.  bash tools/r06_stream.sh outdir
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06/stream}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -12 $OUT/kernel_trace_stats.txt | cut -c1-140
rm -rf $OUT/trace
MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > $OUT/trace_lone_bench.json 2>> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats_lone.txt 2>&1; head -12 $OUT/kernel_trace_stats_lone.txt | cut -c1-140
rm -rf $OUT/trace
MPA_TRACE=1 MPA_TIMING=1 python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/gantt_bench.json 2> $OUT/gantt.err
python tools/gantt.py $OUT/gantt.err > $OUT/pipeline_gantt.txt 2>&1; head -18 $OUT/pipeline_gantt.txt
{ python tools/timing_agg.py < $OUT/gantt.err; grep "mpa-cpu" $OUT/gantt.err | tail -12; } > $OUT/stage_clocks.txt
python tools/bench_line.py $OUT/gantt_bench.json
