#!/bin/bash
# Run tools/scale_run.py --load under several environment settings against ONE saved index (one GPU-box call).
#   tools/sweep.sh <prefix> <outdir> "<scale_run args>" "ENV=V ENV=V|extra args" ...
prefix=$1; out=$2; base=$3; shift 3
mkdir -p "$out"
k=0
for spec in "$@"; do
  envs=${spec%%|*}; extra=""
  [[ "$spec" == *"|"* ]] && extra=${spec#*|}
  echo "=== [$k] $envs | $extra" | tee -a "$out/sweep.log"
  ( env $envs timeout 600 python tools/scale_run.py --load "$prefix" $base $extra ) > "$out/run$k.out" 2> "$out/run$k.err"
  grep -E "OURS|pass|BYTES|DIFFERS|REF" "$out/run$k.out" "$out/run$k.err" | tee -a "$out/sweep.log"
  k=$((k+1))
done
