#!/bin/bash
# k_chain_extract with dynamic LDS (85 VGPRs instead of 166): kernel trace alone and in the stream, before (libmpamd_r06b.so) and after; stream A/B
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_extract; mkdir -p $OUT
WL=/dev/shm/mpa_wl
OLD=$PWD/miniprot_amd/libmpamd_r06b.so
bash tools/r06_ab.sh $OUT "MPA_LIB_PATH=$OLD" "MPA_X=1" 2>&1 | tee $OUT/ab.txt
for v in old new; do
  [ $v = old ] && export MPA_LIB_PATH=$OLD || unset MPA_LIB_PATH
  rm -rf $OUT/trace; MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > /dev/null 2> $OUT/trace.err
  DB=$(find $OUT/trace -name "t_results.db" | head -1)
  python tools/rocpd_stats.py $DB > $OUT/lone_$v.txt 2>&1; echo "== lone $v"; head -6 $OUT/lone_$v.txt | cut -c1-150
  rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 5 --warmup 1 --workload-cache $WL > /dev/null 2> $OUT/trace.err
  DB=$(find $OUT/trace -name "t_results.db" | head -1)
  python tools/rocpd_stats.py $DB > $OUT/stream_$v.txt 2>&1; echo "== stream $v"; head -8 $OUT/stream_$v.txt | cut -c1-150
  rm -rf $OUT/trace
done
unset MPA_LIB_PATH
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
