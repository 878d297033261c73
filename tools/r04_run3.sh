#!/bin/bash
# round 4, GPU call 3: GPU tests, ns/row row sweep vs anti-diagonal (fixed prototype), the bench with two DP rounds per batch
# (speculative terminal-exon repeats), lane / mini-batch variants, extraction phase profile
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
{
echo "# ns/row: row sweep (ext_narrow<32>, production) vs anti-diagonal prototype (dp_antidiag.hip), right-extension calls"
echo "# lone wave: 4 calls = one wave; saturated: 16384 calls = 4096 waves (4 per SIMD)"
for AD in 0 1; do
  python tools/time_wide.py 32 20000 4 4 $AD | tail -1
  python tools/time_wide.py 24 20000 4 4 $AD | tail -1
  python tools/time_wide.py 32 4000 16384 4 $AD | tail -1
done
} > $O/antidiag_ns_per_row.txt 2>&1
cat $O/antidiag_ns_per_row.txt
W=/dev/shm/mpa_wl
python bench.py --steps 10 --warmup 2 --workload-cache $W > $O/bench.json 2> $O/bench.err
python tools/bench_line.py $O/bench.json
MPA_DP_LANES=4 python bench.py --steps 8 --warmup 2 --workload-cache $W --no-extra --no-cpu-baseline --no-cli > $O/bench_l4.json 2> $O/bench_l4.err
python tools/bench_line.py $O/bench_l4.json lanes4
MPA_DP_LANES=3 python bench.py --steps 8 --warmup 2 --workload-cache $W --no-extra --no-cpu-baseline --no-cli > $O/bench_l3.json 2> $O/bench_l3.err
python tools/bench_line.py $O/bench_l3.json lanes3
python bench.py --steps 8 --warmup 2 --workload-cache $W --no-extra --no-cpu-baseline --no-cli --batch 10000 > $O/bench_b10000.json 2> $O/bench_b10000.err
python tools/bench_line.py $O/bench_b10000.json b10000
MPA_EXTRACT_PROF=1 MPA_SEEDERS=1 python bench.py --steps 1 --warmup 1 --workload-cache $W --no-extra --no-cpu-baseline --no-cli > $O/bench_prof.json 2> $O/bench_prof.err
grep "extract-prof" $O/bench_prof.err | tail -8
rm -rf $W
