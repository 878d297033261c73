#!/bin/bash
# PMC passes for the extension kernels (run on the GPU box through gpurun): one counter per pass, kernel trace only.
#   bash tools/pmc_round.sh   ->  gpurun_out/pmc_fetch/, gpurun_out/pmc_write/, gpurun_out/pmc_latest.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CMD="python bench.py --no-stream --steps 2 --warmup 1 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
	d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
	rm -rf $d
	timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- $CMD > $d.json 2> $d.err
done
python tools/pmc_summary.py gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db > gpurun_out/pmc_all.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_all.json"))
ext = {k: v for k, v in d.items() if "k_ext" in k}
n = sum(v["launches"] for v in ext.values()) // 1
fetch = sum(v.get("FETCH_SIZE", 0) for v in ext.values())
write = sum(v.get("WRITE_SIZE", 0) for v in ext.values())
# launches are counted once per pass in the merged dict (same kernels in both passes)
out = {"k_ext_hbm_bytes_per_launch": (fetch * 2 + write) * 1024 / n, "launches": n, "fetch_kb_raw": fetch, "write_kb_raw": write,
       "correction": "FETCH_SIZE x2 (gfx950 rocprofv3 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE uncorrected; KB units",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_round.sh) over `bench.py --no-stream --steps 2 --warmup 1`, all k_ext<G> and k_ext_wide<NW> launches",
       "per_kernel": {k: {"launches": v["launches"], "fetch_kb_raw": v.get("FETCH_SIZE"), "write_kb_raw": v.get("WRITE_SIZE")} for k, v in ext.items()}}
json.dump(out, open("gpurun_out/pmc_latest.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
