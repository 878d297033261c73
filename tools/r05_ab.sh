#!/bin/bash
# Round 5 A/B runs on the GPU box (through gpurun): bench.py on a cached config-3 workload under the environment variants given
# as arguments ("NAME:VAR=VAL,VAR=VAL" ...), one summary block per variant.
#   bash tools/r05_ab.sh OUTDIR STEPS WARMUP "base:" "pool768:MPA_DP_POOL=1" "pool512:MPA_DP_POOL=1,MPA_DP_WORKERS=512"
# (the runs recorded in profiles/r05_experiments.txt were made while the pool was the default: there "MPA_DP_POOL=0" marks the launches)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$1; STEPS=$2; WARM=$3; shift 3
mkdir -p $OUT
WL=/dev/shm/mpa_wl
for spec in "$@"; do
	name=${spec%%:*}; vars=${spec#*:}
	envs=$(echo "$vars" | tr ',' ' ')
	( env $envs timeout 600 python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra --no-cli --workload-cache $WL ${BENCH_ARGS:-} ) > $OUT/$name.json 2> $OUT/$name.err
	echo "## $name ($vars) rc=$?"
	python tools/bench_line.py $OUT/$name.json "   $name"
	grep -h "error\|Error\|rror:" $OUT/$name.err | head -3
done
