"""Whole-path parity at SCALE on the GPU, against the reference binary run live on the same box (oracle/_ref/miniprot travels
with the repo).  These are the regimes the small golden cases never reach: `-I` at hundreds of Mbp with N runs and tandem
paralogs, mini-batches big enough that the library switches GPU seeding and the GPU refinement scan on BY ITSELF (thresholds of
host_map.cpp left at their defaults), `-G 500000` with 250 000-row extension windows, and the widest extension classes
(k_ext_wide<8>, k_ext_wide<16>).  About two minutes in total."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import miniprot_amd as mpa
import refbind
import golden  # noqa: F401  (puts tools/ on sys.path)
import gen_synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(refbind.REF_BIN), reason="oracle/_ref/miniprot not present")]
NCPU = os.cpu_count() or 4


@pytest.fixture(scope="module")
def ctx():
    c = mpa.Context(0)
    yield c
    c.close()


def _reference(tmp_path, idx, prots, names, flags):
    mpi, faa = str(tmp_path / "g.mpi"), str(tmp_path / "p.faa")
    idx.dump(mpi)
    gen_synth.write_fasta_aa(faa, prots, names)
    out = subprocess.run([refbind.REF_BIN, "-t%d" % NCPU] + flags + [mpi, faa], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    os.remove(mpi)
    return out


def _first_difference(ours, ref):
    for k, (x, y) in enumerate(zip(ours.split(b"\n"), ref.split(b"\n"))):
        if x != y:
            return "line %d\n ours %r\n ref  %r" % (k, x[:400], y[:400])
    return "lengths differ: %d vs %d bytes" % (len(ours), len(ref))


def test_genome_scale_batch_with_default_thresholds(ctx, tmp_path, monkeypatch, capfd):
    """360 Mbp in 3 contigs (5 % N runs, 2 % tandem paralogs) x 3 300 proteins, -I -u, as mini-batches of 3 000 and 300 through
    mpa_map_batches: the big one has more than 8 M anchors and 100 M refinement-window bases, so both device stages engage on
    their own (checked from the library's timing notes), the small one stays on the host paths, and every output byte equals
    the reference's."""
    for v in ("MPA_GPU_SEED", "MPA_GPU_REFINE"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("MPA_TIMING", "1")
    # k_seed_sift cuts a query into segments of ~49 k anchors, which only queries against a 3 Gbp genome have; here (~7 k anchors
    # per query) the segment size is lowered so that every query is swept by four or five workgroups side by side, as there
    monkeypatch.setenv("MPA_SIFT_SEG", "1500")
    contigs, prots, names = gen_synth.generate(360_000_000, 3, 3300, 5, mu=8.0, sigma=1.6, imax=190000, n_frac=0.05, paralog_frac=0.02)
    idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(3)])
    del contigs
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, min(64, NCPU)))
    idx.to_device(ctx)
    mo = mpa.default_mapopt()
    mo.flag |= 4
    mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
    assert mo.max_intron > 60000
    batches = [mpa.Queries(prots[a:b], names[a:b]) for a, b in ((0, 3000), (3000, 3300))]
    ours = b"".join(mpa.map_batches(ctx, idx, mo, batches, min(64, NCPU)))
    notes = capfd.readouterr().err
    assert "seeding on the GPU" in notes and ("refinement on the GPU" in notes or "refinement scan on the GPU" in notes), "the device stages did not switch on by themselves"
    ref = _reference(tmp_path, idx, prots, names, ["-I", "-u"])
    assert ours == ref, _first_difference(ours, ref)
    assert ours.count(b"\n") >= 3300
    idx.close()


def _chimeras(rng, planted, tails):
    """proteins whose seeds cover one end only: a planted protein's first (or last) 140 residues + a random tail, so that the
    extension call at the other end is `tail` columns wide"""
    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    out = []
    for k, tail in enumerate(tails):
        p = bytes(planted[k % len(planted)])
        junk = aa[rng.integers(0, 20, tail)].tobytes()
        out.append(p[:140] + junk if k % 2 == 0 else junk + p[-140:])
    return out


def test_long_introns_and_the_widest_extension_classes(ctx, tmp_path):
    """-G 500000 (max_intron = bw = 500 000: extension windows of 250 000 rows, align.c:260-261) on a 60 Mbp contig with
    long-intron genes (>= 4 exons, introns to 450 kb), plus chimeric proteins whose unanchored ends are 300..1000 residues
    wide: extension calls of every wide class (k_ext_wide<2> .. <16>) at full window length.  Bytes equal the reference's."""
    contigs, prots, names, planted = gen_synth.generate(60_000_000, 1, 60, 9, mu=10.5, sigma=1.2, imin=1000, imax=450000, min_exons=4, return_planted=True)
    rng = np.random.default_rng(99)
    chim = _chimeras(rng, planted, [100, 100, 200, 230, 300, 400, 500, 520, 600, 800, 900, 1000])
    prots = list(prots) + chim
    names = list(names) + ["chim%02d" % k for k in range(len(chim))]
    idx = mpa.Index.from_nt4(contigs, ["chr1"])
    del contigs
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, min(32, NCPU)))
    idx.to_device(ctx)
    mo = mpa.default_mapopt()
    mo.flag |= 4
    mo.max_intron = mo.bw = 500000
    ours, results = mpa.map_batches(ctx, idx, mo, [mpa.Queries(prots, names)], min(64, NCPU), keep_results=True)
    st = ctx.dp_stats(total=True)
    assert st["n_ext"] > 100 and st["cells_ext"] > 1.5e9           # the long windows were really swept
    ref = _reference(tmp_path, idx, prots, names, ["-G", "500000", "-u"])
    assert ours[0] == ref, _first_difference(ours[0], ref)
    idx.close()


def test_gene_families_with_thirty_secondaries(ctx, tmp_path):
    """The repeat-rich regime a single planted locus per protein never reaches ("Hard parts"): gene FAMILIES --
    12 diverged copies (2-9 % substitutions, some with a one-base deletion, i.e. a frameshift) of each of 16 two-exon genes, scattered
    over both strands of two contigs -- mapped with `-N 30 --outn=30 --outs=0.3 -p 0.3 -u`, so that hit.c's secondary selection (mp_select_sub,
    mp_select_multi_exon, hit.c:159-236), parent assignment and the output filters (map.c:298-305) see a dozen near-equal regions
    per query, chains of equal score compete for shared anchors and every region is refined, extended and traced back.  Bytes equal
    the reference's, run live."""
    codon = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTG", "M": "ATG",
             "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    contigs, prots, names = gen_synth.generate(40_000_000, 2, 300, 17, mu=7.0, sigma=1.2, imax=20000)
    rng = np.random.default_rng(171)
    fam_prots, fam_names = [], []
    for f in range(16):
        p = bytes(prots[f]).decode()
        cds = np.array([code[c] for a in p for c in codon.get(a, "GCT")], np.uint8)
        cut = 3 * int(rng.integers(len(p) // 3, 2 * len(p) // 3)) + int(rng.integers(0, 3))      # intron in any phase
        ilen = int(rng.integers(200, 6000))
        intron = rng.integers(0, 4, ilen).astype(np.uint8)
        intron[:2] = [2, 3]; intron[-2:] = [0, 2]; intron[2] = 0; intron[-3] = 1                 # GTA ... CAG
        gene = np.concatenate([cds[:cut], intron, cds[cut:], np.array([3, 0, 0], np.uint8)])
        for k in range(12):
            cp = gene.copy()
            m = rng.random(len(cp)) < rng.uniform(0.02, 0.09)
            cp[m] = rng.integers(0, 4, int(m.sum())).astype(np.uint8)
            if k % 4 == 3:
                cp = np.delete(cp, int(rng.integers(30, len(cp) - 30)))
            c = int(rng.integers(0, 2))
            at = int(rng.integers(1000, len(contigs[c]) - len(cp) - 1000))
            contigs[c][at:at + len(cp)] = comp[cp[::-1]] if rng.random() < 0.5 else cp
        fam_prots.append(bytes(prots[f])), fam_names.append("fam%02d" % f)
    prots = list(prots[16:]) + fam_prots
    names = list(names[16:]) + fam_names
    idx = mpa.Index.from_nt4(contigs, ["chr1", "chr2"])
    del contigs
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, min(32, NCPU)))
    idx.to_device(ctx)
    mo = mpa.default_mapopt()
    mo.flag |= 4
    mo.best_n = 30
    mo.out_n = 30
    mo.out_sim = 0.3                                            # --outs: secondaries down to 30 % of the best score are printed
    mo.pri_ratio = 0.3                                          # -p: ... and kept
    n = len(prots)
    batches = [mpa.Queries(prots[a:b], names[a:b]) for a, b in ((0, n // 2), (n // 2, n))]
    ours = b"".join(mpa.map_batches(ctx, idx, mo, batches, min(64, NCPU)))
    ref = _reference(tmp_path, idx, prots, names, ["-N", "30", "--outn=30", "--outs=0.3", "-p", "0.3", "-u"])
    assert ours == ref, _first_difference(ours, ref)
    fam_lines = [l for l in ours.split(b"\n") if l.startswith(b"fam")]
    assert len(fam_lines) >= 16 * 8, "the families did not produce their secondaries (%d lines)" % len(fam_lines)
    idx.close()


def test_a_contig_longer_than_2_31_bases(ctx, tmp_path):
    """One contig of 2.4 Gbp (the reference carries contig offsets and positions as 64-bit integers: ntseq.c, NEWS.md "contigs
    longer than 2 Gbp"): genes planted on both sides of base 2^31, the k-mer table built on the GPU, the .mpi dumped for the
    reference to restore.  Every byte equals the reference's, and hits do lie beyond 2^31."""
    import miniprot_amd.synth as synth
    rng = np.random.default_rng(2031)
    L, piece = 2_400_000_000, 300_000_000
    g = np.concatenate([rng.choice(4, size=piece, p=[0.295, 0.205, 0.205, 0.295]).astype(np.uint8) for _ in range(L // piece)])
    prots, names = [], []
    starts = [100_000_000, 900_000_000, 2_000_000_000, 2_147_400_000, 2_147_483_000, 2_150_000_000, 2_200_000_000, 2_300_000_000, 2_390_000_000]
    for k, st in enumerate(starts * 3):
        st += 30_000_000 * (k // len(starts))
        if st + 400_000 > L:
            st = L - 500_000 - 1000 * k
        prot, gene = synth.make_gene(rng, int(max(80, rng.normal(380, 120))), 8.0, 1.5, 70, 60000, 1)
        rev = rng.random() < 0.5
        seg = synth.COMP[gene[::-1]] if rev else gene
        g[st:st + len(seg)] = seg
        prots.append(synth.mutate(rng, prot))
        names.append("q%02d_%s_%d" % (k, "-" if rev else "+", st))
    idx = mpa.Index.from_nt4([g], ["chrBig"])
    del g
    assert idx.build_kmers(n_threads=min(64, NCPU), ctx=ctx) in ("gpu", "host")
    idx.to_device(ctx)
    mo = mpa.default_mapopt()
    mo.flag |= 4
    mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
    ours = b"".join(mpa.map_batches(ctx, idx, mo, [mpa.Queries(prots, names)], min(32, NCPU)))
    ref = _reference(tmp_path, idx, prots, names, ["-I", "-u"])
    assert ours == ref, _first_difference(ours, ref)
    beyond = [l for l in ours.split(b"\n") if l and l.split(b"\t")[5] == b"chrBig" and int(l.split(b"\t")[7]) > (1 << 31)]
    assert len(beyond) >= 8, "no hit beyond base 2^31"
    idx.close()
