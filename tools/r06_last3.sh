#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_last3; mkdir -p $OUT
(time python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log | cut -c1-120
bash tools/prof_round6.sh $OUT bench trace clocks 2>&1 | tail -40
