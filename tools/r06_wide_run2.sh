#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_wide /tmp/wl
python bench.py --gpus 1 --steps 20 --warmup 5 --workload-cache /tmp/wl > gpurun_out/r06_wide/bench.json 2> gpurun_out/r06_wide/bench.err; tail -3 gpurun_out/r06_wide/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_wide/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['paf_identical'], d['hbm_resident_gb'], d['cli_end_to_end']['wall_s'], d['config5']['value'], d['config5']['paf_identical'], d['config2']['value'])
print(d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['rounds_in_flight_avg'], d['dp_wall_ms_per_batch'], d['seeding_wall_ms_per_batch'])
PY
MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 python bench.py --gpus 1 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-cli --workload-cache /tmp/wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lone', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
