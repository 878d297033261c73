#!/usr/bin/env python3
"""Per-stage timeline of mpa_map_batches() from the MPA_TRACE=1 lines on stderr: busy time per stage, and an ASCII Gantt chart.
    MPA_TRACE=1 python tools/scale_run.py ... 2> log ; python tools/gantt.py log [last-pass-only]"""
import re, sys
ev = []
for l in open(sys.argv[1]):
    m = re.match(r"\[mpa-trace\]\s+([0-9.]+) (\S+)\s+batch\s+(\d+) (begin|end)", l)
    if m:
        ev.append((float(m.group(1)), m.group(2), int(m.group(3)), m.group(4)))
# passes restart the clock: keep the last one
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] < ev[i - 1][0] - 50:
        cut = i
ev = ev[cut:]
spans, open_ = [], {}
for t, st, b, w in ev:
    if w == "begin":
        open_[(st, b)] = t
    else:
        spans.append((st, b, open_.pop((st, b)), t))
if not spans:
    sys.exit("no trace lines")
T = max(s[3] for s in spans)
W = 120
for st in sorted(set(s[0] for s in spans)):
    row = [" "] * W
    busy = 0
    for s in spans:
        if s[0] != st:
            continue
        busy += s[3] - s[2]
        a, b = int(s[2] / T * (W - 1)), max(int(s[3] / T * (W - 1)), int(s[2] / T * (W - 1)))
        for x in range(a, b + 1):
            row[x] = "0123456789abcdefghijklmnopqrstuvwxyz"[s[1] % 36]
    print("%-7s|%s| busy %7.1f ms (%4.1f%%)" % (st, "".join(row), busy, 100 * busy / T))
print("total %.1f ms" % T)
