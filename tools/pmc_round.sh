#!/bin/bash
# PMC passes over the DP kernels (run on the GPU box through gpurun): counters only with --kernel-trace, one block of counters
# per pass, as MI355X_MICROARCH.md prescribes (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass).
#   bash tools/pmc_round.sh [outdir]   ->  <outdir>/pmc_*/ (rocpd databases), <outdir>/pmc_summary.json
# Workload: config 3 scaled to 300 Mbp x 2 000 proteins (one mini-batch of the headline shape, -I -u), blocking steps so
# that every launch is one round of one batch.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
CMD=${PMC_CMD:-"python bench.py --scale 0.1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload-cache /dev/shm/mpa_wl"}
declare -A PASS
PASS[sq]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
PASS[lds]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
for p in sq lds fetch write; do
	rm -rf $OUT/pmc_$p
	MPA_DP_LANES=${PMC_LANES:-1} MPA_SEEDERS=1 MPA_PLANNERS=1 timeout 900 rocprofv3 --pmc ${PASS[$p]} --kernel-trace -d $OUT/pmc_$p -o p -- $CMD > $OUT/pmc_$p.json 2> $OUT/pmc_$p.err
done
python tools/pmc_summary.py $(find $OUT -name "p_results.db" | sort) > $OUT/pmc_all.json
python tools/pmc_report.py $OUT/pmc_all.json > $OUT/pmc_summary.json
head -c 1500 $OUT/pmc_summary.json
