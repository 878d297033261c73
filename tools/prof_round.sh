set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; tail -1 gpurun_out/bench_r01.json | cut -c1-400
python bench.py --no-stream --steps 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r01_blocking.json; cut -c1-220 gpurun_out/bench_r01_blocking.json
export TMPDIR=/tmp
rm -rf gpurun_out/trace_final; rocprofv3 --kernel-trace --stats -d gpurun_out/trace_final -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/trace_final.json 2> gpurun_out/trace_final.err
ls gpurun_out/trace_final | head
