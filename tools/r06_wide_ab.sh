#!/bin/bash
# old (previous commit's library, miniprot_amd/libmpamd_r06a.so) vs new wide extension rows: ns per row, config 3 stream, config 5
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_wide_ab; mkdir -p $OUT
OLD=$PWD/miniprot_amd/libmpamd_r06a.so
{
for al in 200 500 1000; do
  echo -n "old: "; MPA_LIB_PATH=$OLD python tools/time_wide.py $al 20000 2 4 | tail -1
  echo -n "new: "; python tools/time_wide.py $al 20000 2 4 | tail -1
done
} > $OUT/ns_per_row.txt 2>&1
cat $OUT/ns_per_row.txt
bash tools/r06_ab.sh $OUT "MPA_LIB_PATH=$OLD" "MPA_X=1" "MPA_DP_LANES=6" "MPA_DP_LANES=6,MPA_SEEDERS=4" 2>&1 | tee $OUT/ab.txt
for rep in 1 2; do
  for v in "MPA_LIB_PATH=$OLD" "MPA_X=1"; do
    env $v python bench.py --config 5 --no-cpu-baseline --no-cli --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 $v rep $rep', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done 2>&1 | tee $OUT/config5.txt
