#!/bin/bash
# Round 6 evidence on the GPU box (through gpurun): the bench line as the driver runs it, the GPU tests, smoke, kernel traces (stream
# and near-lone), pipeline timeline and stage clocks, ns per row and saturated launches, PMC passes at FULL config 3, the cold start of
# the relinked command line, and the N = 2 code path on one GPU.   bash tools/prof_round6.sh [outdir] [what...]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06_final}
shift
WHAT=${*:-bench tests trace clocks nsrow pmc cold n2}
mkdir -p $OUT
WL=/dev/shm/mpa_wl
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has bench; then
  (time python bench.py --gpus 1 --steps 20 --warmup 5 --workload-cache $WL) > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py $OUT/bench.json "bench (driver settings)"
else
  python bench.py --no-extra --no-cpu-baseline --no-cli --steps 1 --warmup 1 --workload-cache $WL > /dev/null 2>&1
fi
if has tests; then
  (time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-160
fi
if has trace; then
  rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/trace_bench.json 2> $OUT/trace.err
  DB=$(find $OUT/trace -name "t_results.db" | head -1)
  python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats.txt 2>&1; head -9 $OUT/kernel_trace_stats.txt | cut -c1-140
  python tools/timeline.py $DB 2500 60 > $OUT/dispatch_timeline.txt 2>&1
  rm -rf $OUT/trace
  MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > $OUT/trace_lone_bench.json 2>> $OUT/trace.err
  DB=$(find $OUT/trace -name "t_results.db" | head -1)
  python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats_lone.txt 2>&1
  rm -rf $OUT/trace
fi
if has clocks; then
  MPA_TRACE=1 MPA_TIMING=1 MPA_DP_TOP=1 python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > $OUT/gantt_bench.json 2> $OUT/gantt.err
  python tools/gantt.py $OUT/gantt.err > $OUT/pipeline_gantt.txt 2>&1; head -16 $OUT/pipeline_gantt.txt
  { python tools/timing_agg.py < $OUT/gantt.err; grep "mpa-cpu" $OUT/gantt.err | tail -12; grep "mpa-pools" $OUT/gantt.err | tail -13; } > $OUT/stage_clocks.txt
  grep "mpa-dp-top" $OUT/gantt.err | tail -4 > $OUT/dp_round_top_units.txt
fi
if has nsrow; then
  bash tools/r06_nsrow.sh > $OUT/dp_ns_per_row.txt 2>&1
  { echo "# saturated launches, narrow extension, 32 columns x 4000 rows: 12288 calls = 3 waves per SIMD (default: 3 workgroups per CU), 16384 calls = 4 waves per SIMD (MPA_DP_WG_PER_CU=4)";
    python tools/time_wide.py 32 4000 12288 4 | tail -1; MPA_DP_WG_PER_CU=4 python tools/time_wide.py 32 4000 16384 4 | tail -1;
    python tools/time_wide.py 64 4000 6144 4 | tail -1; MPA_DP_WG_PER_CU=4 python tools/time_wide.py 64 4000 8192 4 | tail -1; } >> $OUT/dp_ns_per_row.txt 2>&1
  { echo "# wide extension classes (ext_wide_body), 20 000 rows, one pair of calls: 200 columns = one workgroup, 500 = two, 1000 = four; then 256 pairs of 1000 columns x 4000 rows";
    for al in 200 500 1000; do python tools/time_wide.py $al 20000 2 4 | tail -1; done; python tools/time_wide.py 1000 4000 256 4 | tail -1; } >> $OUT/dp_ns_per_row.txt 2>&1
  cat $OUT/dp_ns_per_row.txt
  python tools/lite_check.py 2>&1 | head -1 > $OUT/checkpointed_traceback_blocks.txt; cat $OUT/checkpointed_traceback_blocks.txt
fi
if has pmc; then
  export PMC_CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL"
  export PMC_WORKLOAD_NOTE="config 3 at FULL size (3 Gbp x 20 000 proteins, -I -u, 4 000-protein mini-batches, two DP rounds per batch), 2 passes over the workload, MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1"
  bash tools/pmc_round.sh $OUT/pmc > $OUT/pmc.log 2>&1
  find $OUT/pmc -name "*.db" -delete; find $OUT/pmc -type d -empty -delete
  python - $OUT/pmc/pmc_summary.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for kn in ("k_dp_round","k_walk","k_chain_extract","k_seed_sift<4096>","k_prep_rows","k_refine_scan_map"):
        k=d["kernels"].get(kn,{})
        print("PMC", kn, {x:(round(k[x],4) if isinstance(k.get(x),float) else k.get(x)) for x in ("avg_us","hbm_bytes_per_launch","hbm_gb_per_s","avg_resident_waves_per_simd","valu_issue_share_of_wave_cycles","wait_any_share","lds_bank_conflict_rate")})
except Exception as e: print("pmc summary:", repr(e)[:200])
PY
fi
if has cold; then bash tools/r05_cold.sh $OUT/cold > /dev/null 2>&1; cp $OUT/cold/cli_cold_start.txt $OUT/cli_cold_start.txt; tail -12 $OUT/cli_cold_start.txt; fi
if has n2; then
  for SC in weak strong; do
    ( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --debug-shared-gpu --scaling $SC --batch 1000 --steps 1 --warmup 1 --no-extra --no-cli --threads 8 --workload-cache $WL ) > $OUT/bench_n2_shared_gpu_debug_$SC.json 2> $OUT/n2_$SC.err
    python tools/bench_line.py $OUT/bench_n2_shared_gpu_debug_$SC.json "N=2 shared GPU (debug, $SC; bench.py spawned its own ranks)"
    tail -2 $OUT/n2_$SC.err | cut -c1-200
  done
fi
bash tools/hostinfo.sh > $OUT/hostinfo.txt 2>&1; head -3 $OUT/hostinfo.txt
rm -f $OUT/trace.err $OUT/pmc/*.err
du -sh $OUT
