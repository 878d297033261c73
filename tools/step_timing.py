#!/usr/bin/env python3
"""Stage-level timing of one mapping step (MPA_TIMING=1 prints the library's own stage clocks).
    MPA_TIMING=1 python tools/step_timing.py [threads] [genome_mb] [n_ctg] [n_prot] [set_I]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import miniprot_amd as mpa, gen_synth
from hostpipe import map_batch_gpu
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gmb = float(sys.argv[2]) if len(sys.argv) > 2 else 50
nctg = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nprot = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
set_I = len(sys.argv) > 5 and sys.argv[5] == "1"
t0 = time.time()
contigs, prots, names = gen_synth.generate(int(gmb * 1e6), nctg, nprot, 12)
t1 = time.time()
idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(nctg)]); mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, max(8, threads)))
t2 = time.time()
q = mpa.Queries(prots, names); mo = mpa.default_mapopt(); mo.flag |= 4
if set_I: mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
ctx = mpa.Context(0); idx.to_device(ctx)
print("generate %.1fs index %.1fs upload %.1fs; max_intron %d" % (t1 - t0, t2 - t1, time.time() - t2, mo.max_intron), file=sys.stderr)
for it in range(3):
    t0 = time.time(); paf = map_batch_gpu(ctx, idx, mo, q, threads)
    dt = time.time() - t0
    res = sum(int(l.split(b"\t")[3]) - int(l.split(b"\t")[2]) for l in paf.split(b"\n") if l and l.split(b"\t")[5] != b"*")
    print("step %.1f ms  %.2f M residues/s" % (dt * 1e3, res / dt / 1e6), file=sys.stderr)
