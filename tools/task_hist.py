"""Shapes of the DP calls of round 1 (columns, rows, share of rows and cells per kernel class) for a synthetic batch; CPU only.
    python tools/task_hist.py"""
import sys, ctypes as C, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import miniprot_amd as mpa, gen_synth
contigs, prots, names = gen_synth.generate(20000000, 1, 400, 12)
idx = mpa.Index.from_nt4(contigs, ["chr1"]); mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 8))
q = mpa.Queries(prots, names); mo = mpa.default_mapopt(); mo.flag |= 4
L = mpa.lib(); L.mpa_batch_begin.restype = C.c_void_p
b = L.mpa_batch_begin(idx.h, C.byref(mo), C.byref(q.c), 8)
ptr = C.c_void_p(); dp = mpa.DpOpt(); n = L.mpa_batch_dp_tasks(C.c_void_p(b), C.byref(ptr), C.byref(dp))
t = np.frombuffer((C.c_char * (n * mpa.DP_TASK.itemsize)).from_address(ptr.value), dtype=mpa.DP_TASK)
ext = t[(t["flag"] & 6) != 0]; glob = t[(t["flag"] & 6) == 0]
print("round-1 tasks", n, "ext", len(ext), "glob", len(glob))
for name, x in (("ext", ext), ("glob", glob)):
    al = x["al"]; nl = x["nl"].astype(np.int64)
    cells = nl * ((al + 7) // 8 * 8)
    print(name, "al percentiles", np.percentile(al, [10, 25, 50, 75, 90, 99]).astype(int), "nl pct", np.percentile(nl, [10, 50, 90, 99]).astype(int))
    for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 64), (65, 128), (129, 256), (257, 1024), (1025, 100000)):
        m = (al >= lo) & (al <= hi)
        print("   al %5d-%-6d tasks %5d (%4.1f%%) rows %5.1f%% cells %5.1f%%" % (lo, hi, m.sum(), 100 * m.mean(), 100 * nl[m].sum() / nl.sum(), 100 * cells[m].sum() / cells.sum()))
