#!/bin/bash
# k_walk launched per class with that class's LDS (8 / 11 / 17 / 29 KB instead of 29 KB for all): tests, then stream A/B against the build before it
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_walk; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/r06_ab.sh $OUT "MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_r06c.so" "MPA_X=1" 2>&1 | tee $OUT/ab.txt
