#!/bin/bash
# round 4, GPU call 1: the GPU test suite, the bench under the new protocol (one step = one pass over the 20 000 proteins), then a
# few stream variants on the cached workload (mini-batch size, extraction phase profile)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
W=/dev/shm/mpa_wl
python bench.py --steps 8 --warmup 2 --workload-cache $W > $O/bench.json 2> $O/bench.err
python tools/bench_line.py $O/bench.json 2>/dev/null | head -40
for B in 6667 10000; do
  python bench.py --steps 6 --warmup 2 --workload-cache $W --no-extra --no-cpu-baseline --batch $B > $O/bench_b$B.json 2> $O/bench_b$B.err
  python tools/bench_line.py $O/bench_b$B.json 2>/dev/null | head -30
done
MPA_EXTRACT_PROF=1 MPA_TIMING=1 python bench.py --steps 1 --warmup 1 --workload-cache $W --no-extra --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
grep -E "extract-prof|seeding on the GPU|seed: |refine: |stage A|mpa_dp_run|pool growth" $O/bench_prof.err | sort | uniq -c | sort -rn | head -5
grep "extract-prof" $O/bench_prof.err | tail -12
rm -rf $W
