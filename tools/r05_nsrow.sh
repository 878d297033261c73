#!/bin/bash
# ns per row of the narrow extension sweep, lone wave and saturated machine, worker pool against one launch per round
cd "${GRAFT_REPO_ROOT:-.}"
for POOL in 0 1; do
  echo "# MPA_DP_POOL=$POOL"
  MPA_DP_POOL=$POOL python tools/time_wide.py 32 20000 4 4 | tail -1
  MPA_DP_POOL=$POOL python tools/time_wide.py 32 4000 16384 4 | tail -1
  MPA_DP_POOL=$POOL python tools/time_wide.py 32 4000 4096 1 | tail -1
done
