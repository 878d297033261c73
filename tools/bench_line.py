#!/usr/bin/env python3
"""One-line summary of a bench.py JSON record: python tools/bench_line.py <label> <file>"""
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M res/s", round(d["ms_per_step"], 1), "ms/step", d.get("md5_first_pass"), d.get("paf_identical"),
          "host_cpu_s/step", round(d.get("host_cpu_s_per_step") or 0, 3), "throttled", d.get("host_throttled"))
except Exception as e:
    print(sys.argv[1], "NO RECORD:", repr(e)[:200])
