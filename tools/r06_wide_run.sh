#!/bin/bash
# ns per row of the wide extension classes (one call of `al` columns per workgroup / per split group, lone), then the DP parity tests
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_wide
{
for al in 200 256 500 1000; do
  python tools/time_wide.py $al 20000 2 4 | tail -1
done
python tools/time_wide.py 1000 4000 256 4 | tail -1
python tools/time_wide.py 200 4000 1536 4 | tail -1
} > gpurun_out/r06_wide/ns_per_row.txt 2>&1
cat gpurun_out/r06_wide/ns_per_row.txt
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | tail -5
timeout 600 python tools/fuzz_dp.py 150 5000 wide > gpurun_out/r06_wide/fuzz_wide.txt 2>&1; tail -4 gpurun_out/r06_wide/fuzz_wide.txt
timeout 400 python tools/fuzz_dp.py 90 7000 > gpurun_out/r06_wide/fuzz.txt 2>&1; tail -2 gpurun_out/r06_wide/fuzz.txt
