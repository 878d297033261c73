#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_last; mkdir -p $OUT
./tools/ubench/corun > $OUT/corun.txt 2>&1; cat $OUT/corun.txt
for k in 1 2 3; do python tools/time_wide.py 200 20000 2 4 | tail -3; done 2>&1 | tee $OUT/wide200.txt
bash tools/prof_round6.sh $OUT bench 2>&1 | tail -12
