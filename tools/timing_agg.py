"""Aggregate MPA_TIMING=1 stderr lines ("[mpa-timing] label  x ms") into count / mean / total per label.
python tools/timing_agg.py < stderr.txt"""
import sys, re, collections
tot, cnt = collections.OrderedDict(), collections.Counter()
for l in sys.stdin:
    m = re.match(r"\[mpa-timing\] (.*?)\s+([0-9.]+) ms", l)
    if m:
        k = m.group(1)
        tot[k] = tot.get(k, 0.0) + float(m.group(2)); cnt[k] += 1
for k, v in tot.items():
    print("%-50s n=%5d mean %9.3f ms total %10.1f ms" % (k, cnt[k], v / cnt[k], v))
