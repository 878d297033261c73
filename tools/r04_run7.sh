#!/bin/bash
# round 4, GPU call 7: tests after the refinement-scan change; config 5 at larger mini-batches; the bench with the 8 000-protein extra
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
grep -h "passed\|failed" $O/pytest_gpu.log | tail -2
W=/dev/shm/mpa_wl
for B in 500 1000 2000; do
  python bench.py --config 5 --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-cli --workload-cache $W > $O/c5_b$B.json 2> $O/c5_b$B.err
  python tools/bench_line.py $O/c5_b$B.json "config5 batch $B"
done
python bench.py --steps 20 --warmup 5 --workload-cache $W > $O/bench.json 2> $O/bench.err
python tools/bench_line.py $O/bench.json "bench (driver settings)"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("b8000:", d.get("config3_batch8000"))
PY
rm -f $O/*.err
rm -rf $W
