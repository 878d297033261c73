#!/bin/bash
# Round-end evidence on the GPU box (through gpurun): tests, smoke, the bench line, kernel trace of the bench command, PMC passes.
#   bash tools/prof_round.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/round}
mkdir -p $OUT
(time python bench.py --workload-cache /dev/shm/mpa_wl) > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
(time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-200
rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 6 --workload-cache /dev/shm/mpa_wl > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -12 $OUT/kernel_trace_stats.txt
python tools/timeline.py $DB 2500 60 > $OUT/dispatch_timeline.txt 2>&1
rm -rf $OUT/trace                      # (only the summaries travel back: gpurun returns at most 64 MiB)
bash tools/pmc_round.sh $OUT/pmc
find $OUT/pmc -name "*.db" -delete; find $OUT/pmc -type d -empty -delete
du -sh $OUT
