#!/bin/bash
# Round 5 evidence on the GPU box (through gpurun): the bench line as the driver runs it, the GPU tests, smoke, kernel traces (stream
# and near-lone), pipeline timeline and stage clocks, PMC passes at FULL config 3, the cold start of the relinked command line with
# its timing notes, the DP worker pool next to the launches (bench line, unit trace, one saturated launch), and the N = 2 code
# path on one GPU.   bash tools/prof_round5.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r05_final}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
(time python bench.py --steps 20 --warmup 5 --workload-cache $WL) > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py $OUT/bench.json "bench (driver settings)"
(time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-100
# cold start of the drop-in, with the library's timing notes (index restore, uploads, pool growth, first batches)
if [ -x examples/_relink/miniprot ] && [ -f $WL/c3/g.mpi ]; then
  ( export LD_LIBRARY_PATH=$PWD/miniprot_amd:$LD_LIBRARY_PATH; /usr/bin/time -v env MPA_TIMING=1 examples/_relink/miniprot -t16 -I -u $WL/c3/g.mpi $WL/c3/p.faa > /tmp/cli.paf ) 2> $OUT/cli_cold_timing.err
  { echo "# examples/_relink/miniprot -t16 -I -u g.mpi p.faa (config 3, 20 000 proteins), cold, MPA_TIMING=1: the reference's own progress lines carry wall clock @ CPU share"; grep -E "^\[M::|index upload|pool growth|Elapsed|Maximum resident" $OUT/cli_cold_timing.err | head -60; python tools/timing_agg.py < $OUT/cli_cold_timing.err 2>/dev/null | grep -E "index upload|pool growth|seeding on the GPU|mpa_dp_run|stage A" ; md5sum /tmp/cli.paf; } > $OUT/cli_cold_start.txt 2>&1
  tail -5 $OUT/cli_cold_start.txt
fi
# kernel trace of the bench command (shorter job)
rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -9 $OUT/kernel_trace_stats.txt | cut -c1-140
python tools/timeline.py $DB 2500 60 > $OUT/dispatch_timeline.txt 2>&1
rm -rf $OUT/trace
# the same kernels nearly alone
rm -rf $OUT/trace; MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > $OUT/trace_lone_bench.json 2>> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats_lone.txt 2>&1
rm -rf $OUT/trace
# pipeline timeline + stage clocks
MPA_TRACE=1 MPA_TIMING=1 python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/gantt_bench.json 2> $OUT/gantt.err
python tools/gantt.py $OUT/gantt.err > $OUT/pipeline_gantt.txt 2>&1; head -18 $OUT/pipeline_gantt.txt
{ python tools/timing_agg.py < $OUT/gantt.err; grep "mpa-cpu" $OUT/gantt.err | tail -12; } > $OUT/stage_clocks.txt
# the DP worker pool next to the launches: bench line, unit trace (resident waves over time), kernel trace, one saturated launch
MPA_DP_POOL=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL > $OUT/bench_pool.json 2> $OUT/bench_pool.err; python tools/bench_line.py $OUT/bench_pool.json "worker pool (MPA_DP_POOL=1)"
rm -f /tmp/dptrace.txt; MPA_DP_POOL=1 MPA_DP_TRACE=/tmp/dptrace.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL > /dev/null 2>> $OUT/bench_pool.err
python tools/dp_trace.py /tmp/dptrace.txt 5 > $OUT/dp_pool_unit_trace.txt 2>&1; tail -2 $OUT/dp_pool_unit_trace.txt
rm -rf $OUT/trace; MPA_DP_POOL=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 3 --warmup 1 --workload-cache $WL > /dev/null 2>> $OUT/trace.err
python tools/rocpd_stats.py $(find $OUT/trace -name "t_results.db" | head -1) > $OUT/kernel_trace_stats_pool.txt 2>&1; head -4 $OUT/kernel_trace_stats_pool.txt | cut -c1-140
rm -rf $OUT/trace
bash tools/r05_sat.sh $OUT/sat 2>&1 | grep -v "prep_rows\|k_dp_arm\|l2_writeback" | sed 's/W2026.*finalization.*//' > $OUT/dp_saturated_launch.txt; rm -rf $OUT/sat
# PMC at full config 3, one lane / seeder / planner (counters serialise the kernels anyway)
export PMC_CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL"
export PMC_WORKLOAD_NOTE="config 3 at FULL size (3 Gbp x 20 000 proteins, -I -u, 4 000-protein mini-batches, two DP rounds per batch), 2 passes over the workload, MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1"
bash tools/pmc_round.sh $OUT/pmc > $OUT/pmc.log 2>&1
find $OUT/pmc -name "*.db" -delete; find $OUT/pmc -type d -empty -delete
python - $OUT/pmc/pmc_summary.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for kn in ("k_dp_round","k_chain_extract","k_seed_sift<4096>","k_prep_rows","k_refine_scan_map"):
        k=d["kernels"].get(kn,{})
        print("PMC", kn, {x:(round(k[x],4) if isinstance(k.get(x),float) else k.get(x)) for x in ("avg_us","hbm_bytes_per_launch","hbm_gb_per_s","avg_resident_waves_per_simd","valu_issue_share_of_wave_cycles","lds_bank_conflict_rate")})
except Exception as e: print("pmc summary:", repr(e)[:200])
PY
# the N = 2 code path with both ranks on this one GPU (gloo; NOT a scaling number): identity at N > 1, dynamic deal, both scaling modes
for SC in weak strong; do
  ( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --debug-shared-gpu --scaling $SC --batch 1000 --steps 1 --warmup 1 --no-extra --no-cli --threads 8 --workload-cache $WL ) > $OUT/bench_n2_shared_gpu_debug_$SC.json 2> $OUT/n2_$SC.err
  python tools/bench_line.py $OUT/bench_n2_shared_gpu_debug_$SC.json "N=2 shared GPU (debug, $SC)"
  tail -2 $OUT/n2_$SC.err | cut -c1-200
done
bash tools/hostinfo.sh > $OUT/hostinfo.txt 2>&1; head -3 $OUT/hostinfo.txt
rm -f $OUT/*.err $OUT/pmc/*.err
rm -rf $WL
du -sh $OUT
