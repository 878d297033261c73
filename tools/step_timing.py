import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tools")
import miniprot_amd as mpa, gen_synth
from hostpipe import *
t0=time.time()
contigs, prots, names = gen_synth.generate(50000000, 1, 1000, 12)
idx = mpa.Index.from_nt4(contigs, ["chr1"]); mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 8))
print("setup", time.time()-t0)
q = mpa.Queries(prots, names); mo = mpa.default_mapopt(); mo.flag |= 4
ctx = mpa.Context(0); idx.to_device(ctx)
for it in range(3):
    t0=time.time(); paf = map_batch_gpu(ctx, idx, mo, q, int(sys.argv[1]) if len(sys.argv)>1 else 64); print("step %.1f ms" % ((time.time()-t0)*1e3), file=sys.stderr)
