#!/usr/bin/env python3
"""Summary of a bench.py JSON record: python tools/bench_line.py <file> [label]"""
import json, sys
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    r = d.get("roofline") or {}
    print(label, "%.2f M res/s" % (d["value"] / 1e6), "%.1f ms/step" % d["ms_per_step"], "%.1f ms/mini-batch" % (d.get("ms_per_mini_batch") or 0),
          "batch", (d.get("config") or {}).get("proteins_per_mini_batch"), "md5", (d.get("md5_first_pass") or "")[:8], "identical", d.get("paf_identical"))
    print("   frac %.4f  launch %.1f ms x %s  rounds_in_flight %.2f  wall-frac %.4f" % (r.get("frac") or 0, r.get("avg_launch_ms") or 0, r.get("launches"),
          d.get("rounds_in_flight_avg") or 0, (r.get("wall") or {}).get("frac") or 0))
    print("   stage wall ms/batch: sketch %s seed %s plan %s dp %s out %s | hbm %.1f GB  growths timed %s total %s | host cpu-s/mini-batch %.2f throttled %s" % tuple(
          [("%.0f" % d[k]) if d.get(k) is not None else "-" for k in ("sketch_wall_ms_per_batch", "seeding_wall_ms_per_batch", "planning_wall_ms_per_batch", "dp_wall_ms_per_batch", "output_wall_ms_per_batch")] +
          [d.get("hbm_resident_gb") or 0, d.get("pool_growth_events_timed"), d.get("pool_growth_events_total"), d.get("host_cpu_s_per_mini_batch") or d.get("host_cpu_s_per_step") or 0, (d.get("host_throttled") or {}).get("periods")]))
    if d.get("cli_end_to_end"):
        print("   cli:", d["cli_end_to_end"])
    if d.get("cpu_baseline"):
        print("   cpu: %.3f M res/s on %s threads" % (d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
    for k in ("config2", "config5"):
        if k in d:
            e = d[k]
            print("  ", k, e if "error" in e else "%.2f M res/s %.1f ms/step identical %s frac %s" % (e["value"] / 1e6, e["ms_per_step"], e["paf_identical"], e.get("roofline_frac")))
except Exception as e:
    print(label, "NO RECORD:", repr(e)[:200])
