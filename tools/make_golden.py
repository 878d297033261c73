#!/usr/bin/env python3
"""Regenerate tests/golden/ from the REAL reference (oracle/_ref, compiled from /root/reference by
oracle/Makefile).  Run in the authoring container only; the outputs are committed.

  * DPP3 fixture (the reference's own test data) and the reference's PAF for it (config 1)
  * reference PAF for the synthetic cases of tests/golden.py (inputs are regenerated from their seeds)
  * dp_vectors.npz: random DP calls with the reference's ns_global_gs16b() answers (all three modes)
  * gs32_vectors.npz: calls of the 32-bit operator (ns_global_gs32b, -msse4.1 build of the reference): `make_golden.py gs32` makes only these
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


def make_gs32_vectors():
    """tests/golden/gs32_vectors.npz: calls of the 32-bit operator with the answers of the reference's ns_global_gs32b in its
    -msse4.1 build (oracle/_ref/libnasw_sse41.so; oracle/Makefile and miniprot_amd/csrc/gs32_core.h say why that build)."""
    from gs32util import gs32_cases
    rng = np.random.default_rng(3232)
    nts, aas, sss, flags, pars, mats, scores, cigs = [], [], [], [], [], [], [], []
    for nt, aa, P, flag, ss in gs32_cases(rng, 180):
        r = refbind.ref_gs32(nt, aa, P, flag, ss)
        nts.append(np.frombuffer(nt, np.uint8)), aas.append(np.frombuffer(aa, np.uint8))
        sss.append(np.frombuffer(bytes(ss), np.uint8) if ss is not None else np.zeros(0, np.uint8))
        flags.append(flag), pars.append([P.go, P.ge, P.io, P.fs] + list(P.sp)), mats.append(P.mat.copy())
        scores.append(r[2]), cigs.append(np.array(r[3], np.uint32))
    np.savez_compressed(golden.path("gs32_vectors.npz"), nt=np.concatenate(nts), nt_len=np.array([len(x) for x in nts]),
                        aa=np.concatenate(aas), aa_len=np.array([len(x) for x in aas]), ss=np.concatenate(sss), ss_len=np.array([len(x) for x in sss]),
                        flag=np.array(flags), par=np.array(pars, np.int64), mat=np.array(mats, np.int8), score=np.array(scores, np.int64),
                        cig=np.concatenate(cigs), cig_len=np.array([len(x) for x in cigs]))
    print("gs32_vectors", len(flags), "calls,", sum(1 for x in scores if x > 32767), "with scores beyond int16")


def main():
    refbind.build_oracle()
    if len(sys.argv) > 1 and sys.argv[1] == "gs32":
        make_gs32_vectors()
        return
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
    make_gs32_vectors()


if __name__ == "__main__":
    main()
