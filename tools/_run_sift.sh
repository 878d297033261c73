cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04_sift; mkdir -p $OUT; WL=/dev/shm/mpa_wl
(time python -m pytest tests/test_seed_gpu.py tests/test_scale_gpu.py -m gpu -q -x) > $OUT/pytest_seed.log 2>&1; grep -h "passed\|failed" $OUT/pytest_seed.log | tail -1
rm -rf $OUT/trace; MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > $OUT/trace_lone_bench.json 2>> $OUT/trace.err
DB=$(find $OUT/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_trace_stats_lone.txt 2>&1; grep -h "k_seed_sift\|k_dp_round" $OUT/kernel_trace_stats_lone.txt | cut -c1-150
rm -rf $OUT/trace
python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline --no-cli --workload-cache $WL > $OUT/bench2.json 2> $OUT/bench2.err; python tools/bench_line.py $OUT/bench2.json "bench"
