#!/usr/bin/env python3
"""Resident DP units over time from an MPA_DP_TRACE file (one line per unit: slot generation unit kind priority start end, 100-MHz
device ticks; written by the DP worker pool, dp_exec.hip).
    python tools/dp_trace.py trace.txt [bin_ms]
Prints, per time bin: how many units were running (workgroup units count as 4 waves, one-wave units as 1), how many rounds had a
unit running, and the share of the running waves that belonged to units longer than 10 ms (the stragglers); then the totals:
wave-seconds of DP work, the time during which any unit ran, the mean number of busy waves while the pool ran."""
import sys
import collections

KIND_WAVES = {4: 4, 5: 4, 11: 4}          # U_EXT_W4, U_EXT_SPLIT, U_GLOB_W4 take a workgroup; every other kind is one wave


def main():
    path = sys.argv[1]
    bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    ev = []
    for line in open(path):
        f = line.split()
        if len(f) < 7:
            continue
        slot, gen, unit, kind, prio, t0, t1 = (int(x) for x in f[:7])
        if t0 <= 0 or t1 <= t0:
            continue
        ev.append((t0, t1, KIND_WAVES.get(kind, 1), (slot, gen)))
    if not ev:
        print("no units in", path)
        return
    base = min(e[0] for e in ev)
    end = max(e[1] for e in ev)
    tick_ms = 1e-5                                             # 100 MHz
    n_bins = int((end - base) * tick_ms / bin_ms) + 1
    waves = [0.0] * n_bins
    strag = [0.0] * n_bins
    rounds = [set() for _ in range(n_bins)]
    wave_ms = 0.0
    durs = []
    for t0, t1, w, rid in ev:
        a, b = (t0 - base) * tick_ms, (t1 - base) * tick_ms
        wave_ms += (b - a) * w
        durs.append(b - a)
        k0, k1 = int(a / bin_ms), int(b / bin_ms)
        for k in range(k0, k1 + 1):
            lo, hi = max(a, k * bin_ms), min(b, (k + 1) * bin_ms)
            if hi > lo:
                waves[k] += (hi - lo) / bin_ms * w
                if b - a > 10.0:
                    strag[k] += (hi - lo) / bin_ms * w
                rounds[k].add(rid)
    busy_bins = sum(1 for x in waves if x > 0)
    print("# bin %.1f ms: time  busy waves  rounds with a unit running  share of the busy waves in units > 10 ms" % bin_ms)
    for k in range(n_bins):
        if k < 400:
            print("%8.1f %10.1f %4d %6.2f" % (k * bin_ms, waves[k], len(rounds[k]), strag[k] / waves[k] if waves[k] else 0.0))
    durs.sort()
    n = len(durs)
    print("# units %d; unit ms p50 %.3f p90 %.3f p99 %.3f max %.3f" % (n, durs[n // 2], durs[n * 9 // 10], durs[n * 99 // 100], durs[-1]))
    print("# wave-ms of DP work %.1f; span %.1f ms; bins with a unit running %d of %d; mean busy waves while running %.1f (4 096 = four per SIMD)" % (
        wave_ms, (end - base) * tick_ms, busy_bins, n_bins, wave_ms / max(busy_bins * bin_ms, 1e-9)))


if __name__ == "__main__":
    main()
