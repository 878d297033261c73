#!/bin/bash
# What the GPU box gives the host side: visible CPUs, cgroup quota, NUMA, and a busy-loop scaling probe.
echo "nproc: $(nproc)  online: $(cat /sys/devices/system/cpu/online)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period: $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null) $(cat /sys/fs/cgroup/cpuset/cpuset.cpus 2>/dev/null)"
grep -m1 "model name" /proc/cpuinfo; lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA|MHz" | head -12
free -g | head -2
cat /proc/loadavg
python3 - <<'PY'
import multiprocessing as mp, time, os
def spin(q):
    t0 = time.time(); n = 0
    while time.time() - t0 < 1.0:
        for _ in range(10000): n += 1
    q.put(n)
for k in (1, 8, 16, 32, 64, 128, 256):
    q = mp.Queue(); ps = [mp.Process(target=spin, args=(q,)) for _ in range(k)]
    t0 = time.time(); [p.start() for p in ps]; tot = sum(q.get() for _ in ps); [p.join() for p in ps]
    print("procs %3d: %6.1f M iters/s aggregate, wall %.2f s" % (k, tot / 1e6, time.time() - t0))
PY
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
