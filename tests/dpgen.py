"""Deterministic random DP tasks (protein vs. genomic window) exercising every state of the
spliced/frameshift DP: introns in all three phases with canonical and non-canonical signals,
substitutions, amino-acid indels, frameshifts, in-frame stops, N runs, al<8, al%8==0, long flanks."""
import numpy as np

AA = b"ARNDCQEGHILKMFPSTWYV"
# standard genetic code, codon index = n1<<4|n2<<2|n3 with A0 C1 G2 T3
_CODE = {}
_first = "TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG"
_second = "TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG"
_third = "TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG"
_amino = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_N = {"A": 0, "C": 1, "G": 2, "T": 3}
for a, b, c, m in zip(_first, _second, _third, _amino):
    _CODE.setdefault(m, []).append((_N[a], _N[b], _N[c]))


def back_translate(prot, rng):
    out = []
    for ch in prot:
        cods = _CODE[chr(ch)]
        out.extend(cods[rng.integers(len(cods))])
    return out


def make_task(rng, al=None, max_intron=400, flank=60, p_intron=0.08, p_sub=0.12, p_indel=0.02, p_fs=0.01, p_n=0.002):
    if al is None:
        al = int(rng.choice([rng.integers(1, 9), rng.integers(8, 41), rng.integers(8, 41), rng.integers(40, 130)]))
    prot = bytes(AA[i] for i in rng.integers(0, 20, al))
    nt = []
    for k in range(al):
        cod = list(back_translate(prot[k:k + 1], rng))
        if rng.random() < p_intron:  # intron inside/after this codon at phase ph
            ph = int(rng.integers(0, 3))
            ilen = int(rng.integers(20, max_intron))
            body = list(rng.integers(0, 4, ilen))
            r = rng.random()
            if r < 0.7:
                body[0:2] = [2, 3]
                body[2] = int(rng.choice([0, 2]))
                body[-2:] = [0, 2]
                body[-3] = int(rng.choice([1, 3]))
            elif r < 0.8:
                body[0:2] = [2, 1]
                body[-2:] = [0, 2]
            elif r < 0.9:
                body[0:2] = [0, 3]
                body[-2:] = [0, 1]
            cod = cod[:ph] + body + cod[ph:]
        if rng.random() < p_fs:
            if rng.random() < 0.5:
                cod = cod[:-1]
            else:
                cod = cod + [int(rng.integers(0, 4))]
        nt.extend(cod)
    # the query diverges from the planted protein
    q = bytearray()
    for ch in prot:
        r = rng.random()
        if r < p_sub:
            q.append(AA[rng.integers(0, 20)])
        elif r < p_sub + p_indel:
            continue
        elif r < p_sub + 2 * p_indel:
            q.append(ch)
            q.append(AA[rng.integers(0, 20)])
        else:
            q.append(ch)
    if len(q) == 0:
        q.append(AA[0])
    if rng.random() < 0.05:
        q[int(rng.integers(0, len(q)))] = ord("*")
    if rng.random() < 0.05:
        q[int(rng.integers(0, len(q)))] = ord("X")
    lf = int(rng.integers(0, flank))
    rf = int(rng.integers(0, flank))
    nt = list(rng.integers(0, 4, lf)) + nt + list(rng.integers(0, 4, rf))
    nt = np.array(nt, dtype=np.uint8)
    if len(nt) < 3:
        nt = np.concatenate([nt, rng.integers(0, 4, 3).astype(np.uint8)])
    mask = rng.random(len(nt)) < p_n
    nt[mask] = 4
    return bytes(nt), bytes(q)


def make_ss(rng, nl, density=0.02):
    """A synthetic splice-score track in the reference's encoding (ntseq.c:234-296): 0xff = unset,
    else score+64 << 1 | is_acceptor."""
    ss = np.full(nl, 0xff, dtype=np.uint8)
    idx = np.nonzero(rng.random(nl) < density)[0]
    sc = rng.integers(-10, 15, len(idx))
    acc = rng.integers(0, 2, len(idx))
    ss[idx] = ((sc + 64) << 1 | acc).astype(np.uint8)
    return bytes(ss)
