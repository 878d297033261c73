#!/usr/bin/env python3
"""Derived per-kernel metrics from the merged PMC dump of tools/pmc_summary.py (gfx950 corrections of MI355X_MICROARCH.md applied
where the guide gives one): HBM bytes per launch, LDS bank-conflict rate, VALU busy, wave residency.
    python tools/pmc_report.py pmc_all.json > pmc_summary.json"""
import hashlib, json, re, sys
d = json.load(open(sys.argv[1]))
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def dp_kernel_source_md5():
    """identity of the DP kernels' source (what bench.py compares before it quotes these counters next to a run)"""
    h = hashlib.md5()
    for f in ("dp_kernels.hip", "dp_device.h"):
        h.update(open(os.path.join(ROOT, "miniprot_amd", "csrc", f), "rb").read())
    return h.hexdigest()
out = {"source": "rocprofv3 --pmc <block> --kernel-trace, one pass per counter block (tools/pmc_round.sh); workload: "
                 + os.environ.get("PMC_WORKLOAD_NOTE", "config 3 scaled x0.1 (300 Mbp x 2 000 proteins, -I -u), 3 blocking steps, MPA_DP_LANES=1"),
       "dp_kernel_source_md5": dp_kernel_source_md5(),
       "corrections": "FETCH_SIZE x2 (gfx950 rocprofv3 tallies 128-B requests as 64 B, guide section HBM), KB units; WRITE_SIZE uncorrected; SQ_*_CYCLES in quad-cycles", "kernels": {}}
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(ILi(\d+))?", n)
    return (m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")) if m else n[:40]
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("total_us", 0)):
    if "k_" not in k or v.get("total_us", 0) < 50:
        continue
    n = max(v.get("launches", 1), 1)
    r = {"launches": n, "avg_us": v["total_us"] / n}
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        r["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / n
        r["hbm_gb_per_s"] = r["hbm_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
    if v.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_rate"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]      # extra LDS cycles / all LDS-array cycles
        r["lds_insts_per_launch"] = v.get("SQ_INSTS_LDS", 0) / n
    if v.get("SQ_WAVE_CYCLES"):
        r["waves_per_launch"] = v.get("SQ_WAVES", 0) / n
        r["valu_issue_share_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"]
        r["wait_any_share"] = v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"]
        r["wait_inst_any_share"] = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        r["valu_insts_per_launch"] = v.get("SQ_INSTS_VALU", 0) / n
        if v.get("SQ_BUSY_CYCLES"):
            # average resident waves per SQ-busy cycle, over 256 CUs x 4 SIMDs (quad-cycle units cancel)
            r["avg_resident_waves_per_simd"] = v["SQ_WAVE_CYCLES"] / v["SQ_BUSY_CYCLES"] / 4.0
    out["kernels"][short(k)] = r
print(json.dumps(out, indent=1))
