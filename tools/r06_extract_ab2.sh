#!/bin/bash
# k_chain_extract: LDS stage of 1024 / 512 / 256 words (13.3 / 9.2 / 7.2 KB per wave: 3 / 4 / 5 waves per SIMD) -- lone kernel time and stream
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_extract2; mkdir -p $OUT
WL=/dev/shm/mpa_wl
bash tools/r06_ab.sh $OUT "MPA_X=1" "MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_xs512.so" "MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_xs256.so" 2>&1 | tee $OUT/ab.txt
for v in 1024 512 256; do
  [ $v = 1024 ] && unset MPA_LIB_PATH || export MPA_LIB_PATH=$PWD/miniprot_amd/libmpamd_xs$v.so
  rm -rf $OUT/trace; MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 2 --warmup 1 --workload-cache $WL > /dev/null 2> $OUT/trace.err
  DB=$(find $OUT/trace -name "t_results.db" | head -1)
  python tools/rocpd_stats.py $DB > $OUT/lone_$v.txt 2>&1; echo "== lone stage $v"; grep "k_chain_extract" $OUT/lone_$v.txt | cut -c1-150
  rm -rf $OUT/trace
done
