#!/bin/bash
# round 4, GPU call 2: GPU tests (new: anti-diagonal prototype vs oracle, device chains vs the reference's mp_chain, gene families),
# ns/row of the row sweep vs the anti-diagonal prototype, the bench at two mini-batch sizes, extraction phase profile
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
{
echo "# ns/row: row sweep (ext_narrow<32>, production) vs anti-diagonal prototype (dp_antidiag.hip), right-extension calls of 32 columns"
echo "# lone wave: 4 calls = one wave; saturated: 16384 calls = 4096 waves (4 per SIMD)"
for AD in 0 1; do
  python tools/time_wide.py 32 20000 4 4 $AD | tail -1
  python tools/time_wide.py 24 20000 4 4 $AD | tail -1
  python tools/time_wide.py 32 4000 16384 4 $AD | tail -1
done
} > $O/antidiag_ns_per_row.txt 2>&1
cat $O/antidiag_ns_per_row.txt
W=/dev/shm/mpa_wl
python bench.py --steps 8 --warmup 2 --workload-cache $W > $O/bench.json 2> $O/bench.err
python tools/bench_line.py $O/bench.json
for B in 10000; do
  python bench.py --steps 8 --warmup 2 --workload-cache $W --no-extra --no-cpu-baseline --batch $B > $O/bench_b$B.json 2> $O/bench_b$B.err
  python tools/bench_line.py $O/bench_b$B.json
done
MPA_EXTRACT_PROF=1 MPA_SEEDERS=1 python bench.py --steps 1 --warmup 1 --workload-cache $W --no-extra --no-cpu-baseline --no-cli > $O/bench_prof.json 2> $O/bench_prof.err
grep "extract-prof" $O/bench_prof.err | tail -8
rm -rf $W
