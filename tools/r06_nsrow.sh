#!/bin/bash
# ns per row of the narrow sweeps: lone wave and a saturated machine (3 waves per SIMD: 3072 waves), extension (flag 4) and traceback (flag 1;
# MPA_DP_LITE_MIN=0: the plain traceback sweep, default: the checkpointed path)
cd "${GRAFT_REPO_ROOT:-.}"
for al in 16 32 64; do
  python tools/time_wide.py $al 20000 $((256 / al)) 4 | tail -1
  python tools/time_wide.py $al 4000 $((12288 * 32 / al)) 4 | tail -1
done
for lm in 0 384; do
  echo "# MPA_DP_LITE_MIN=$lm"
  MPA_DP_LITE_MIN=$lm python tools/time_wide.py 32 20000 4 1 | tail -1
  MPA_DP_LITE_MIN=$lm python tools/time_wide.py 32 2000 6144 1 | tail -1
  MPA_DP_LITE_MIN=$lm python tools/time_wide.py 64 2000 3072 1 | tail -1
done
