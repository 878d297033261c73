import sys, os, time, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, miniprot_amd as mpa, gen_synth
gmb, nctg = float(sys.argv[1]), int(sys.argv[2])
contigs, prots, names = gen_synth.generate(int(gmb * 1e6), nctg, 10, 3)
t0 = time.time()
idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(nctg)]); t1 = time.time()
mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 128)); t2 = time.time()
idx.dump("/tmp/ours.mpi"); t3 = time.time()
print("ours: pack %.1fs build %.1fs (128 threads) dump %.1fs" % (t1 - t0, t2 - t1, t3 - t2))
gen_synth.write_fasta_nt("/tmp/g.fa", contigs)
t4 = time.time()
subprocess.run(["/root/repo/oracle/_ref/miniprot", "-t256", "-d", "/tmp/ref.mpi", "/tmp/g.fa"], check=True, stderr=subprocess.DEVNULL)
t5 = time.time()
print("reference: miniprot -t256 -d: %.1fs (FASTA parse included)" % (t5 - t4))
import hashlib
def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""): h.update(blk)
    return h.hexdigest()
print("identical .mpi:", md5("/tmp/ours.mpi") == md5("/tmp/ref.mpi"))
