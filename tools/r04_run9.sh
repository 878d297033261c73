#!/bin/bash
# round 4, GPU call 9: mini-batch size at the DRIVER's job length (20 steps = 20 passes over the 20 000 proteins), new stream hand-off test
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_map_gpu.py -m gpu -q -k "handoff" 2>&1 | tail -2
W=/dev/shm/mpa_wl
for B in 4000 6667 10000 5000; do
  python bench.py --steps 20 --warmup 5 --batch $B --no-extra --no-cpu-baseline --no-cli --workload-cache $W > $O/b$B.json 2> $O/b$B.err
  python tools/bench_line.py $O/b$B.json "20 steps, batch $B"
done
rm -f $O/*.err
rm -rf $W
