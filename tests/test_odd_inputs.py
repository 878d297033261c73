"""Edge cases the reference accepts (ragged and degenerate inputs), whole path, diffed against the reference
binary run live (oracle/_ref/miniprot; skipped where it is absent -- the GPU box runs the same inputs through the HIP
executor in tests/test_map_gpu.py::test_odd_inputs_on_gpu, against the same binary, which travels with the repo)."""
import os
import subprocess
import numpy as np
import pytest
import miniprot_amd as mpa
import golden
import gen_synth
from hostpipe import map_batch, oracle_executor

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "miniprot")


def odd_inputs(seed=31, chimera=11):
    """contigs: a normal one with genes, a 90-bp one, an all-N one, one with IUPAC codes and lowercase (as FASTA text);
    proteins: ordinary, lowercase, with X / * / U / B, a single residue, five residues, a 4000-residue chimera, poly-K"""
    rng = np.random.default_rng(seed)
    contigs, prots, names, planted = gen_synth.generate(400000, 1, 12, seed, return_planted=True)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    g0 = lut[contigs[0]].tobytes()
    g0 = g0[:1000] + g0[1000:3000].lower() + g0[3000:]
    iupac = bytearray(lut[rng.integers(0, 4, 3000)].tobytes())
    for k in rng.integers(0, 3000, 60):
        iupac[k] = rng.choice(list(b"RYKMSWBDHVNn"))
    fasta = b">chrA some description\n" + g0 + b"\n>tiny\n" + lut[rng.integers(0, 4, 90)].tobytes() + b"\n>allN\n" + b"N" * 700 + b"\n>iupac\n" + bytes(iupac) + b"\n"
    p = [bytes(x) for x in planted]
    q = list(prots[:8])
    q.append(p[8].lower())
    q.append(p[9][:50] + b"X" + p[9][51:120] + b"*" + p[9][121:200] + b"UB" + p[9][202:])
    q += [b"M", b"MKTAY", b"".join(p[:chimera]), b"K" * 300]
    qn = ["q%02d" % i for i in range(len(q))]
    return fasta, q, qn


def write_inputs(tmp, fasta, q, qn):
    gfa, pfa = os.path.join(tmp, "g.fa"), os.path.join(tmp, "p.fa")
    open(gfa, "wb").write(fasta)
    with open(pfa, "wb") as f:
        for n, s in zip(qn, q):
            f.write(b">" + n.encode() + b"\n" + s + b"\n")
    return gfa, pfa


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/miniprot not built")
@pytest.mark.parametrize("flags", [["-u", "--gff"], ["-u", "-I"]])
def test_odd_inputs_match_live_reference(oracle_built, tmp_path, flags):
    fasta, q, qn = odd_inputs(chimera=4)        # (the oracle executor is slow on very long proteins; the GPU test uses 11)
    gfa, pfa = write_inputs(str(tmp_path), fasta, q, qn)
    ref = subprocess.run([REF, "-t2"] + flags + [gfa, pfa], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    idx = mpa.Index.from_fasta(gfa)
    case = {"flags": flags, "genome": idx.genome_len(), "n_ctg": 1}
    mo = golden.mapopt_for(case)
    if "-I" in flags:
        import ctypes as C
        mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
    ours = golden.file_header(case) + map_batch(idx, mo, mpa.Queries(q, qn), oracle_executor, 2)
    assert ours == ref
    idx.close()
