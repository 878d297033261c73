#!/bin/bash
# run-to-run spread of the driver's command on one box: three times, the main line only (no CPU baseline, no extra configs, no command-line leg)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_repeat; mkdir -p $OUT
WL=/dev/shm/mpa_wl
python bench.py --no-extra --no-cpu-baseline --no-cli --steps 1 --warmup 1 --workload-cache $WL > /dev/null 2>&1
for k in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-cli --workload-cache $WL > $OUT/run$k.json 2> $OUT/run$k.err
  python tools/bench_line.py $OUT/run$k.json "run $k" | head -2
done | tee $OUT/repeat.txt
