#!/bin/bash
# occupancy-limiter counters (SPI resource-allocation stalls) of the kernels, each kernel alone (PMC serialises): what a workgroup of k_dp_round waits for
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_spi; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SPI_RA_[A-Z_0-9]*\|SPI_CSN_[A-Z_0-9]*\|SPI_CS[A-Z_0-9]*\|TCP_UTCL1_[A-Z_0-9]*\|TCC_[A-Z_0-9]*TLB[A-Z_0-9]*\|SQ_LEVEL_WAVES\|SQ_OCCUPANCY[A-Z_0-9]*\|MeanOccupancy[A-Za-z]*\|OccupancyPercent" | sort -u > $OUT/counters_avail.txt
cat $OUT/counters_avail.txt | tr '\n' ' '; echo
WL=/dev/shm/mpa_wl
python bench.py --no-extra --no-cpu-baseline --no-cli --steps 1 --warmup 1 --workload-cache $WL > /dev/null 2>&1
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-cli --workload-cache $WL"
run_pass() { # name counters...
  n=$1; shift
  rm -rf $OUT/pmc_$n
  MPA_DP_LANES=1 MPA_SEEDERS=1 MPA_PLANNERS=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$n -o p -- $CMD > /dev/null 2> $OUT/pmc_$n.err
  python tools/pmc_summary.py $(find $OUT/pmc_$n -name "p_results.db") > $OUT/pmc_$n.json 2>> $OUT/pmc_$n.err
  find $OUT/pmc_$n -name "*.db" -delete; rm -rf $OUT/pmc_$n
}
run_pass ra1 SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN
run_pass ra2 SPI_RA_LDS_CU_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TMP_STALL_CSN
run_pass tlb TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT
ls -la $OUT; head -c 1200 $OUT/pmc_ra1.json; tail -3 $OUT/pmc_ra1.err
