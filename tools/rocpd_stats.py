#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel table and a per-dispatch list.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--dispatches]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    rows = list(cur.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch order by start"))
    agg = {}
    for k, s, e, g, w in rows:
        name = sym.get(k, str(k))
        a = agg.setdefault(name, [0, 0, 0, 1 << 62])
        a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s); a[3] = min(a[3], e - s)
    tot = sum(a[1] for a in agg.values()) or 1
    print("%-70s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, a in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-70s %6d %12.1f %12.1f %12.1f %12.1f %6.1f" % (name[:70], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[3] / 1e3, a[2] / 1e3, 100.0 * a[1] / tot))
    if "--dispatches" in sys.argv:
        t0 = rows[0][1] if rows else 0
        for k, s, e, g, w in rows:
            print("%10.3f ms  +%10.3f ms  grid %7d wg %4d  %s" % ((s - t0) / 1e6, (e - s) / 1e6, g, w, sym.get(k, str(k))[:80]))


if __name__ == "__main__":
    main()
