"""Static resources of every kernel of the library from the code-object metadata of `hipcc -S` (VGPRs, SGPRs, static LDS, scratch, threads per
workgroup) and the waves per SIMD the registers allow (512 VGPRs per SIMD lane, granule 8, at most 8 waves).
python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt   (compiles miniprot_amd/csrc/dp_exec.hip for gfx950: about a minute)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "miniprot_amd", "csrc", "dp_exec.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "dp_exec.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
rows = []
for blk in text.split("  - .agpr_count:")[1:]:
    f = {k: v for k, v in re.findall(r"\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|max_flat_workgroup_size|vgpr_spill_count):\s+(\S+)", blk)}
    if "name" not in f:
        continue
    name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name).replace("mpa::", "")
    if "rocprim" in name or name.startswith("__"):
        continue
    v = int(f["vgpr_count"])
    waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
    rows.append((name, v, int(f["sgpr_count"]), int(f["group_segment_fixed_size"]), int(f["private_segment_fixed_size"]), int(f.get("vgpr_spill_count", 0)), int(f["max_flat_workgroup_size"]), waves))
print("# Static resources of every kernel of libmpamd.so (hipcc --offload-arch=gfx950 -O3, code-object metadata) and the occupancy limit the VGPRs set")
print("# (512 VGPRs per SIMD lane, granule 8, at most 8 waves per SIMD).  Dynamic LDS on top of the static figure: k_dp_round 41 216 B per workgroup")
print("# (3 per CU, MPA_DP_WG_PER_CU), k_walk 7 968 / 11 040 / 17 184 / 29 472 by class, k_chain_extract 7 168 per wave, k_refine_scan_map 8 << hs_log2 + 4 544")
print("# (12.7-37 KB), k_seed_sift static only, k_glob_wide / k_ext_huge as launched.  Next to three resident DP waves (384 VGPRs) a SIMD holds one wave of")
print("# k_seed_sift (128), k_walk (88), two of k_chain_extract (64), one of k_refine_scan_map (72).")
print("%-34s %5s %5s %9s %7s %6s %6s %s" % ("kernel", "VGPR", "SGPR", "LDS(stat)", "scratch", "spills", "WG", "waves/SIMD by VGPR"))
for r in sorted(set(rows)):
    print("%-34s %5d %5d %9d %7d %6d %6d %d" % r)
