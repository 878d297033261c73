#!/usr/bin/env python3
"""Regenerate tests/golden/ from the REAL reference (oracle/_ref, compiled from /root/reference by
oracle/Makefile).  Run in the authoring container only; the outputs are committed.

  * DPP3 fixture (the reference's own test data) and the reference's PAF for it (config 1)
  * reference PAF for the synthetic cases of tests/golden.py (inputs are regenerated from their seeds)
  * dp_vectors.npz: random DP calls with the reference's ns_global_gs16b() answers (all three modes)
  * chain_vectors.npz / sketch vectors: anchor sets with the reference's mp_chain() output
"""
import os
import shutil
import subprocess
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refbind  # noqa: E402
import golden  # noqa: E402
import gen_synth  # noqa: E402
from dpgen import make_task  # noqa: E402

GOLD = golden.GOLD
os.makedirs(GOLD, exist_ok=True)
REF = refbind.REF_BIN


def run_ref(args):
    return subprocess.run([REF, "-t4"] + args, capture_output=True, check=True).stdout


def main():
    refbind.build_oracle()
    # config 1
    for f in ("DPP3-hs.gen.fa.gz", "DPP3-mm.pep.fa.gz"):
        shutil.copy(os.path.join("/root/reference/test", f), os.path.join(GOLD, f))
    open(os.path.join(GOLD, "dpp3.ref.paf"), "wb").write(run_ref([golden.path("DPP3-hs.gen.fa.gz"), golden.path("DPP3-mm.pep.fa.gz")]))
    # synthetic cases
    with tempfile.TemporaryDirectory() as tmp:
        for case in golden.SYNTH_CASES:
            contigs, prots, names = golden.synth_inputs(case)
            fa, faa = os.path.join(tmp, "g.fa"), os.path.join(tmp, "p.fa")
            gen_synth.write_fasta_nt(fa, contigs)
            gen_synth.write_fasta_aa(faa, prots, names)
            extra = ["--spsc=" + golden.write_spsc(case, contigs, os.path.join(tmp, "spsc.tsv"))] if "spsc" in case else []
            out = run_ref(case["flags"] + extra + [fa, faa])
            open(golden.path(case["name"] + ".ref.paf"), "wb").write(out)
            print(case["name"], len(out), "bytes", out.count(b"\n"), "lines")
    # DP vectors
    rng = np.random.default_rng(2024)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    nts, aas, flags, ios, res, cigs = [], [], [], [], [], []
    for k in range(300):
        nt, aa = make_task(rng)
        for fl in (1, 2, 4):
            io = 29 if fl == 1 or rng.random() < 0.7 else 19
            PP = refbind.DpParams(P.mat, io=io)
            r = refbind.ref_nasw(nt, aa, PP, fl)
            nts.append(np.frombuffer(nt, np.uint8)), aas.append(np.frombuffer(aa, np.uint8)), flags.append(fl), ios.append(io)
            res.append(r[:3]), cigs.append(np.array(r[3], np.uint32))
    np.savez_compressed(golden.path("dp_vectors.npz"), nt=np.concatenate(nts), nt_len=np.array([len(x) for x in nts]),
                        aa=np.concatenate(aas), aa_len=np.array([len(x) for x in aas]), flag=np.array(flags), io=np.array(ios),
                        res=np.array(res, np.int64), cig=np.concatenate(cigs), cig_len=np.array([len(x) for x in cigs]), mat=P.mat)
    print("dp_vectors", len(flags))


if __name__ == "__main__":
    main()
