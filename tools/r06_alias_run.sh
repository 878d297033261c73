set -x
mkdir -p gpurun_out/r06_alias
./tools/ubench/corun > gpurun_out/r06_alias/corun.txt 2>&1; cat gpurun_out/r06_alias/corun.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_alias/bench.json 2> gpurun_out/r06_alias/bench.err; tail -3 gpurun_out/r06_alias/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_alias/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['paf_identical'], d['hbm_resident_gb'], d['cli_end_to_end']['wall_s'], d['config5']['value'], d['config2']['value'])
PY
