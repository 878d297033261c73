#!/usr/bin/env python3
"""Whole-path run at an arbitrary scale: synthetic genome -> index -> stream of mini-batches
through mpa_map_batches() -> text; optionally the same job with the reference binary (oracle/_ref/miniprot) for a byte
comparison and a CPU wall time.

    python tools/scale_run.py --genome-mb 3000 --n-ctg 24 --n-prot 20000 --batch 4000 -I --ref
"""
import argparse, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import numpy as np
import miniprot_amd as mpa, gen_synth


def aligned_residues(txt):
    n = 0
    for l in txt.split(b"\n"):
        f = l.split(b"\t")
        if len(f) > 5 and f[5] != b"*" and not l.startswith(b"#"):
            n += int(f[3]) - int(f[2])
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=200)
    ap.add_argument("--n-ctg", type=int, default=4)
    ap.add_argument("--n-prot", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=4000, help="proteins per mini-batch")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("-I", dest="set_I", action="store_true")
    ap.add_argument("-G", dest="set_G", type=int, default=0, help="max intron size (miniprot -G: max_intron = bw = G)")
    ap.add_argument("--min-exons", type=int, default=1)
    ap.add_argument("--imin", type=int, default=70)
    ap.add_argument("--n-frac", type=float, default=0.0)
    ap.add_argument("--paralog-frac", type=float, default=0.0)
    ap.add_argument("--mu", type=float, default=7.5)
    ap.add_argument("--sigma", type=float, default=1.5)
    ap.add_argument("--imax", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--passes", type=int, default=2, help="timed passes over the whole protein set (best is reported)")
    ap.add_argument("--ref", action="store_true", help="also run oracle/_ref/miniprot on the same .mpi and compare bytes")
    ap.add_argument("--no-gpu", action="store_true", help="(debug) skip our mapper")
    ap.add_argument("--tmp", default="/tmp/mpa_scale")
    ap.add_argument("--save", default="", help="write <prefix>.mpi and <prefix>.faa (index + proteins) and exit: inputs for --load")
    ap.add_argument("--load", default="", help="skip generation and index build: restore <prefix>.mpi, read <prefix>.faa")
    a = ap.parse_args()
    ncpu = os.cpu_count() or 1
    thr = a.threads or max(1, min(128, ncpu))
    t0 = time.time()
    if a.load:
        idx = mpa.Index.restore(a.load + ".mpi")
        names, prots = [], []
        for line in open(a.load + ".faa", "rb"):
            (names if line.startswith(b">") else prots).append(line[1:].strip().decode() if line.startswith(b">") else line.strip())
        t1 = t2 = time.time()
        contigs = None
    else:
      contigs, prots, names = gen_synth.generate(int(a.genome_mb * 1e6), a.n_ctg, a.n_prot, a.seed, mu=a.mu, sigma=a.sigma, imin=a.imin, imax=a.imax, n_frac=a.n_frac, min_exons=a.min_exons, paralog_frac=a.paralog_frac)
      t1 = time.time()
      idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(a.n_ctg)])
      mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, max(8, thr)))
      t2 = time.time()
    if a.save:
        idx.dump(a.save + ".mpi")
        gen_synth.write_fasta_aa(a.save + ".faa", prots, names)
        print("saved %s.mpi / .faa (generate %.1fs, index %.1fs)" % (a.save, t1 - t0, t2 - t1), file=sys.stderr)
        return
    mo = mpa.default_mapopt(); mo.flag |= 4
    if a.set_I:
        mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
    if a.set_G:
        mo.max_intron = mo.bw = a.set_G
    print("generate %.1fs, index build %.1fs (%d threads), genome %d bp, max_intron %d" % (t1 - t0, t2 - t1, thr, idx.genome_len(), mo.max_intron), file=sys.stderr)
    batches = [mpa.Queries(prots[i:i + a.batch], names[i:i + a.batch]) for i in range(0, len(prots), a.batch)]
    ours = None
    if not a.no_gpu:
        ctx = mpa.Context(0)
        tu = time.time(); idx.to_device(ctx); print("index upload %.1fs" % (time.time() - tu), file=sys.stderr)
        best = 1e30
        for p in range(a.passes):
            tp = time.time()
            texts = mpa.map_batches(ctx, idx, mo, batches, thr)
            dt = time.time() - tp
            best = min(best, dt)
            print("pass %d: %.3f s" % (p, dt), file=sys.stderr)
        ours = b"".join(texts)
        res = aligned_residues(ours)
        print("OURS  %d proteins in %d mini-batches: %.3f s -> %.2f M aligned residues/s (%d residues; md5 %s)" %
              (len(prots), len(batches), best, res / best / 1e6, res, hashlib.md5(ours).hexdigest()))
    if a.ref:
        os.makedirs(a.tmp, exist_ok=True)
        mpi, faa = os.path.join(a.tmp, "g.mpi"), os.path.join(a.tmp, "p.faa")
        td = time.time(); idx.dump(mpi); print("dump .mpi %.1fs (%.1f GB)" % (time.time() - td, os.path.getsize(mpi) / 1e9), file=sys.stderr)
        with open(faa, "w") as f:
            for n, s in zip(names, prots):
                f.write(">%s\n%s\n" % (n, s if isinstance(s, str) else s.decode()))
        open(os.path.join(a.tmp, "empty.faa"), "w").close()
        exe = os.path.join(ROOT, "oracle", "_ref", "miniprot")
        flags = ["-u"] + (["-I"] if a.set_I else []) + (["-G", str(a.set_G)] if a.set_G else [])
        def run(q):
            t = time.time()
            out = subprocess.run([exe, "-t%d" % ncpu] + flags + [mpi, q], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            return time.time() - t, out
        t_load, _ = run(os.path.join(a.tmp, "empty.faa"))
        t_all, ref = run(faa)
        ref = b"".join(l + b"\n" for l in ref.split(b"\n") if l and not l.startswith(b"##gff"))
        res = aligned_residues(ref)
        print("REF   miniprot -t%d: wall %.3f s, index load %.3f s -> %.2f M aligned residues/s mapping-only (md5 %s)" %
              (ncpu, t_all, t_load, res / max(t_all - t_load, 1e-9) / 1e6, hashlib.md5(ref).hexdigest()))
        if ours is not None:
            print("BYTES IDENTICAL" if ours == ref else "OUTPUT DIFFERS (ours %d bytes, ref %d bytes)" % (len(ours), len(ref)))
        for f in (mpi, faa):
            os.remove(f)


if __name__ == "__main__":
    main()
