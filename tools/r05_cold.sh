#!/bin/bash
# cold start of the relinked command line on config 3 (20 000 proteins): whole-process wall clock, main()'s own clock, the
# library's timing notes (uploads, pool growth, first batches, teardown at exit), and the same with MPA_FAST_EXIT=1
cd "${GRAFT_REPO_ROOT:-.}"
OUT=${1:-gpurun_out/r05_cold}; mkdir -p $OUT
WL=/dev/shm/mpa_wl
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL > /dev/null 2>&1
export LD_LIBRARY_PATH=$PWD/miniprot_amd:$LD_LIBRARY_PATH
run() {   # name, env...
  local name=$1; shift
  python3 - "$OUT/$name.err" "$@" <<'PY'
import os, subprocess, sys, time, hashlib
err = sys.argv[1]; env = dict(os.environ)
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); env[k] = v
wl = "/dev/shm/mpa_wl/c3"
t = time.time()
p = subprocess.run(["examples/_relink/miniprot", "-t16", "-I", "-u", wl + "/g.mpi", wl + "/p.faa"], stdout=subprocess.PIPE, stderr=open(err, "w"), env=env)
print("   whole process %.2f s, rc %d, md5 %s" % (time.time() - t, p.returncode, hashlib.md5(p.stdout).hexdigest()))
PY
}
{
echo "# examples/_relink/miniprot -t16 -I -u g.mpi p.faa (config 3: 3 Gbp, 20 000 proteins = 4 mini-batches of the reference's default -K 2M), one MI355X"
for spec in "plain:" "plain_again:" "timing:MPA_TIMING=1" "fast_exit:MPA_FAST_EXIT=1" "fast_exit_again:MPA_FAST_EXIT=1"; do
  name=${spec%%:*}; vars=${spec#*:}
  echo "## $name $vars"
  run $name $vars
  grep -E "Real time" $OUT/$name.err | sed 's/^/   /'
done
echo "## timing notes of the MPA_TIMING=1 run (sums over threads; the stages overlap)"
grep -E "^\[M::worker_pipeline|^\[M::mp_idx_restore" $OUT/timing.err | head -6 | sed 's/^/   /'
python tools/timing_agg.py < $OUT/timing.err 2>/dev/null | grep -E "index upload|pool growth|seeding on the GPU|mpa_dp_run \(total\)|stage A|A1: seeds|refinement on the GPU"
grep "at exit" $OUT/timing.err
} > $OUT/cli_cold_start.txt 2>&1
cat $OUT/cli_cold_start.txt
rm -f $OUT/*.err
