#!/bin/bash
# round 4, GPU call 5 (short): kernel trace of the stream as it is now; round workgroups per CU 5 / 3 against the default 4
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
W=/dev/shm/mpa_wl
run() {   # label, env, bench args
  ( env $2 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --no-cli --workload-cache $W $3 ) > $O/$1.json 2> $O/$1.err
  python tools/bench_line.py $O/$1.json "$1 [$2 $3]"
}
run base "" ""
run wg5 "MPA_ROUND_WG_PER_CU=5" ""
run wg3 "MPA_ROUND_WG_PER_CU=3" ""
run q24 "GPU_MAX_HW_QUEUES=24" ""
run q12 "GPU_MAX_HW_QUEUES=12" ""
rm -rf $O/trace; rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --no-extra --no-cpu-baseline --no-cli --steps 4 --warmup 1 --workload-cache $W > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/trace -name "t_results.db" | head -1)
python tools/rocpd_stats.py $DB > $O/kernel_trace_stats.txt 2>&1; head -24 $O/kernel_trace_stats.txt | cut -c1-150
rm -rf $O/trace $O/*.err
rm -rf $W
