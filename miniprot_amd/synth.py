#!/usr/bin/env python3
"""Deterministic synthetic genome + protein sets for the benchmark configs.

Genome: iid ACGT background (GC 41 %), optional N runs; planted multi-exon genes on both strands with
GT..AG introns of log-normal length; optional tandem pseudo-paralogs (diverged copies of planted genes next to
them: the stressor for secondary hits / best_n, options.c:62); query proteins = planted proteins with substitutions
and indels.

    python -m miniprot_amd.synth --genome-mb 50 --n-prot 1000 --seed 12 --out-prefix /tmp/c2
writes <prefix>.genome.fa and <prefix>.prot.fa  (tools/gen_synth.py is the same command)
"""
import argparse
import numpy as np

AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
_first = "TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG"
_second = "TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG"
_third = "TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG"
_amino = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_N = {"A": 0, "C": 1, "G": 2, "T": 3}
CODONS = {}
for a, b, c, m in zip(_first, _second, _third, _amino):
    CODONS.setdefault(ord(m), []).append((_N[a], _N[b], _N[c]))
STOPS = CODONS[ord("*")]
NT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_gene(rng, length, mu, sigma, imin, imax, min_exons=1):
    prot = AA[rng.integers(0, 20, length)].copy()
    prot[0] = ord("M")
    cds = np.empty(3 * length + 3, dtype=np.uint8)
    for k, ch in enumerate(prot):
        cods = CODONS[int(ch)]
        cds[3 * k:3 * k + 3] = cods[rng.integers(len(cods))]
    cds[3 * length:] = STOPS[rng.integers(len(STOPS))]
    n_exon = int(rng.integers(max(1, min_exons), max(1, length // 60) + 1)) if length // 60 >= max(1, min_exons) else max(1, min_exons)
    cuts = []
    if n_exon > 1:
        cand = np.arange(20, len(cds) - 20)
        if len(cand) >= n_exon - 1:
            cuts = sorted(rng.choice(cand, n_exon - 1, replace=False).tolist())
    parts, prev = [], 0
    for c in cuts:
        parts.append(cds[prev:c])
        ilen = int(np.clip(rng.lognormal(mu, sigma), imin, imax))
        intron = rng.integers(0, 4, ilen).astype(np.uint8)
        intron[0:2] = (2, 3)
        intron[2] = rng.choice([0, 2])
        intron[-2:] = (0, 2)
        intron[-3] = rng.choice([1, 3])
        parts.append(intron)
        prev = c
    parts.append(cds[prev:])
    return prot, np.concatenate(parts)


def mutate(rng, prot, p_sub=0.15, p_del=0.0075, p_ins=0.0075):
    r = rng.random(len(prot))
    out = []
    for ch, x in zip(prot, r):
        if x < p_sub:
            out.append(int(AA[rng.integers(0, 20)]))
        elif x < p_sub + p_del:
            continue
        elif x < p_sub + p_del + p_ins:
            out.append(int(ch))
            out.append(int(AA[rng.integers(0, 20)]))
        else:
            out.append(int(ch))
    return bytes(out) if out else b"M"


def add_paralogs(contigs, genes, slot, frac, seed):
    """Tandem pseudo-paralogs: diverged copies (0.5-6 % substitutions, rare 1-bp deletions = frameshifts; closer than the
    mapper's secondary-to-primary ratio of 0.7 tolerates only for the nearest ones) of planted
    genes, placed next to the original inside the gene's own slot, until `frac` of the genome is such copies.
    Uses its own generator so that the genome/protein stream of generate() is the same with and without paralogs.
    genes: (contig, start, length, slot index).  Returns the number of copies written."""
    rng = np.random.default_rng([seed, 0x9a7a106])
    target = int(frac * sum(len(c) for c in contigs))
    order = rng.permutation(len(genes))
    done = n_copies = 0
    for gi in order:
        if done >= target:
            break
        ci, start, glen, s = genes[gi]
        lo, hi = s * slot + 500, (s + 1) * slot - 500            # the slot's usable range
        src = contigs[ci][start:start + glen].copy()
        right, left = start + glen, start                        # free space after / before the gene and its copies
        for _ in range(int(rng.integers(1, 4))):
            div = rng.uniform(0.005, 0.06)
            cp = src.copy()
            m = rng.random(len(cp)) < div
            cp[m] = rng.integers(0, 4, int(m.sum())).astype(np.uint8)
            cp = cp[rng.random(len(cp)) >= 0.0003]
            gap = int(rng.integers(200, 2000))
            if right + gap + len(cp) <= hi:
                at = right + gap
                right = at + len(cp)
            elif left - gap - len(cp) >= lo:
                at = left - gap - len(cp)
                left = at
            else:
                break
            contigs[ci][at:at + len(cp)] = cp
            done += len(cp)
            n_copies += 1
    return n_copies


def generate(genome_len, n_ctg, n_prot, seed, mu=7.5, sigma=1.5, imin=70, imax=50000, n_frac=0.0, min_exons=1, mean_len=400, sd_len=160, return_planted=False,
             paralog_frac=0.0):
    rng = np.random.default_rng(seed)
    ctg_len = genome_len // n_ctg
    contigs = []
    for _ in range(n_ctg):
        g = rng.choice(4, size=ctg_len, p=[0.295, 0.205, 0.205, 0.295]).astype(np.uint8)
        contigs.append(g)
    if n_frac > 0:
        for g in contigs:
            n_runs = max(1, int(ctg_len * n_frac / 50000))
            for s in rng.integers(0, ctg_len - 50000, n_runs):
                g[s:s + 50000] = 4
    prots, names, planted = [], [], []
    # planted positions: split every contig into equal slots, one gene per slot
    per_ctg = (n_prot + n_ctg - 1) // n_ctg
    slot = ctg_len // max(per_ctg, 1)
    k = 0
    genes = []
    for ci in range(n_ctg):
        for s in range(per_ctg):
            if k >= n_prot:
                break
            length = int(max(60, rng.normal(mean_len, sd_len)))
            prot, gene = make_gene(rng, length, mu, sigma, imin, imax, min_exons)
            if len(gene) + 2000 > slot:       # too long for its slot: shrink introns by regenerating single-exon
                prot, gene = make_gene(rng, length, mu, 0.1, imin, min(imax, max(imin + 1, (slot - 3 * length - 2100) // max(1, length // 60))), 1)
            room = slot - len(gene) - 1000
            start = s * slot + 500 + int(rng.integers(0, max(room, 1)))
            rev = rng.random() < 0.5
            seg = COMP[gene[::-1]] if rev else gene
            contigs[ci][start:start + len(seg)] = seg
            genes.append((ci, start, len(seg), s))
            planted.append(prot)
            prots.append(mutate(rng, prot))
            names.append("p%05d_c%d_%s_%d" % (k, ci, "-" if rev else "+", start))
            k += 1
    if paralog_frac > 0:
        add_paralogs(contigs, genes, slot, paralog_frac, seed)
    if return_planted:
        return contigs, prots, names, planted
    return contigs, prots, names


def write_fasta_nt(path, contigs):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            s = lut[g]
            for p in range(0, len(s), 1 << 24):
                chunk = s[p:p + (1 << 24)]
                # 80-column lines are not needed by either reader; keep long lines to write fast
                f.write(chunk.tobytes())
                f.write(b"\n")


def write_fasta_aa(path, prots, names):
    with open(path, "wb") as f:
        for n, p in zip(names, prots):
            f.write(b">" + n.encode() + b"\n" + p + b"\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=50)
    ap.add_argument("--n-ctg", type=int, default=1)
    ap.add_argument("--n-prot", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=12)
    ap.add_argument("--intron-mu", type=float, default=7.5)
    ap.add_argument("--intron-sigma", type=float, default=1.5)
    ap.add_argument("--intron-min", type=int, default=70)
    ap.add_argument("--intron-max", type=int, default=50000)
    ap.add_argument("--n-frac", type=float, default=0.0)
    ap.add_argument("--min-exons", type=int, default=1)
    ap.add_argument("--paralog-frac", type=float, default=0.0, help="fraction of the genome that is tandem pseudo-paralogs of planted genes")
    ap.add_argument("--out-prefix", required=True)
    a = ap.parse_args()
    contigs, prots, names = generate(int(a.genome_mb * 1e6), a.n_ctg, a.n_prot, a.seed, a.intron_mu, a.intron_sigma, a.intron_min,
                                     a.intron_max, a.n_frac, a.min_exons, paralog_frac=a.paralog_frac)
    write_fasta_nt(a.out_prefix + ".genome.fa", contigs)
    write_fasta_aa(a.out_prefix + ".prot.fa", prots, names)
