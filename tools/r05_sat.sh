#!/bin/bash
# one saturated launch under rocprofv3: worker pool against one launch per round (kernel durations straight from the trace)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r05f}
mkdir -p $OUT
for spec in "base:MPA_DP_POOL=0" "pool1024:MPA_DP_POOL=1,MPA_DP_WORKERS=1024" "pool768:MPA_DP_POOL=1,MPA_DP_WORKERS=768"; do
  name=${spec%%:*}; vars=${spec#*:}
  for shape in "32 4000 16384 4" "32 4000 4096 1" "32 4000 8192 4"; do
    tag=$(echo $shape | tr ' ' '_')
    rm -rf $OUT/tr
    env $(echo $vars | tr "," " ") rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python tools/time_wide.py $shape > $OUT/${name}_$tag.txt 2>&1
    DB=$(find $OUT/tr -name "t_results.db" | head -1)
    echo "## $name ($vars) shape $shape: $(tail -1 $OUT/${name}_$tag.txt)"
    python tools/rocpd_stats.py $DB 2>/dev/null | grep -E "k_dp_|k_prep_rows|k_l2" | cut -c1-150
  done
done
rm -rf $OUT/tr
