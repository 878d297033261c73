#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_last2; mkdir -p $OUT
(time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-120
timeout 900 python tools/fuzz_dp.py 600 6000 wide > $OUT/fuzz_wide.txt 2>&1; tail -3 $OUT/fuzz_wide.txt
bash tools/prof_round6.sh $OUT bench 2>&1 | tail -12
