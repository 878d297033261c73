#!/bin/bash
# k_refine_scan_map launched per size class of the query's k-mer map (13 / 21 / 37 KB of LDS instead of the batch maximum for all): tests, stream A/B
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_refine; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "refine or golden or reference or family or scale or whole or map" 2>&1 | tail -4
bash tools/r06_ab.sh $OUT "MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_r06d.so" "MPA_X=1" 2>&1 | tee $OUT/ab.txt
