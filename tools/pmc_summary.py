#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc rocpd databases per kernel (sum and per-launch mean of every counter).
    python tools/pmc_summary.py db1 [db2 ...]"""
import sqlite3
import sys
import json


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    pmc = {r[0]: r[1] for r in cur.execute("select id, name from rocpd_info_pmc")}
    sym = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    disp = {r[0]: (sym.get(r[1], str(r[1])), r[3] - r[2]) for r in cur.execute("select event_id, kernel_id, start, end from rocpd_kernel_dispatch")}
    out = {}
    seen = {}
    for ev, pid, val in cur.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
        if ev not in disp:
            continue
        k = disp[ev][0]
        d = out.setdefault(k, {})
        d[pmc[pid]] = d.get(pmc[pid], 0) + val
        seen.setdefault(k, set()).add(ev)
    for k in out:
        out[k]["launches"] = len(seen[k])
        out[k]["total_us"] = sum(disp[e][1] for e in seen[k]) / 1e3
    return out


def main():
    merged = {}
    for p in sys.argv[1:]:
        for k, v in load(p).items():
            merged.setdefault(k, {}).update(v)
    print(json.dumps(merged, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
