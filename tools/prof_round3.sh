#!/bin/bash
# Round 3 evidence on the GPU box (through gpurun): the bench line as the driver runs it, the GPU tests, kernel traces (stream and
# near-lone), PMC passes at FULL config 3, the pipeline timeline, and the N = 2 code path on one GPU.   bash tools/prof_round3.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r03_final2}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
bash tools/hostinfo.sh > $OUT/hostinfo.txt 2>&1; head -3 $OUT/hostinfo.txt
(time python bench.py --steps 20 --warmup 5 --workload-cache $WL) > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py "bench (driver settings)" $OUT/bench.json
(time python bench.py --steps 80 --warmup 10 --no-extra --no-cpu-baseline --workload-cache $WL) > $OUT/bench_80steps.json 2>> $OUT/bench.err; python tools/bench_line.py "bench 80 steps" $OUT/bench_80steps.json
(time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-100
# kernel trace of the bench command
rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -8 $OUT/kernel_trace_stats.txt | cut -c1-140
python tools/timeline.py $DB 2500 60 > $OUT/dispatch_timeline.txt 2>&1
rm -rf $OUT/trace
# the same kernels nearly alone
rm -rf $OUT/trace; MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 --workload-cache $WL > $OUT/trace_lone_bench.json 2>> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats_lone.txt 2>&1
rm -rf $OUT/trace
# pipeline timeline
MPA_TRACE=1 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 --workload-cache $WL > $OUT/gantt_bench.json 2> $OUT/gantt.err
python tools/gantt.py $OUT/gantt.err > $OUT/pipeline_gantt.txt 2>&1; head -14 $OUT/pipeline_gantt.txt
MPA_TIMING=1 python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --workload-cache $WL > $OUT/timing_bench.json 2> $OUT/timing.err
python tools/timing_agg.py < $OUT/timing.err > $OUT/stage_clocks.txt; grep "mpa-cpu" $OUT/timing.err | tail -9 >> $OUT/stage_clocks.txt
# PMC at full config 3, one lane / seeder / planner (counters serialise the kernels anyway)
export PMC_CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload-cache $WL"
export PMC_WORKLOAD_NOTE="config 3 at FULL size (3 Gbp x 20 000 proteins, -I -u, 4 000-protein mini-batches), 3 steps, MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1"
bash tools/pmc_round.sh $OUT/pmc > $OUT/pmc.log 2>&1
find $OUT/pmc -name "*.db" -delete; find $OUT/pmc -type d -empty -delete
python - <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03_final2/pmc/pmc_summary.json"))
    k=d["kernels"].get("k_dp_round",{})
    print("PMC k_dp_round:", {x:k.get(x) for x in ("avg_us","hbm_bytes_per_launch","avg_resident_waves_per_simd","valu_issue_share_of_wave_cycles","lds_bank_conflict_rate")})
except Exception as e: print("pmc summary:", repr(e)[:200])
PY
# the N = 2 code path with both ranks on this one GPU (gloo; NOT a scaling number): identity at N > 1, dynamic deal
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --debug-shared-gpu --batch 1000 --steps 10 --warmup 2 --no-extra --threads 8 --workload-cache $WL ) > $OUT/bench_n2_shared_gpu_debug.json 2> $OUT/n2.err
python tools/bench_line.py "N=2 shared GPU (debug)" $OUT/bench_n2_shared_gpu_debug.json
tail -3 $OUT/n2.err | cut -c1-200
rm -f $OUT/*.err
du -sh $OUT
