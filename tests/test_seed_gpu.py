"""Operator-level parity of the GPU seeding stage (seed_exec.hip: anchor gather, (query, block) sort, forward pass of the
pre-chain, compaction; host: chain extraction from the sparse view) against the host implementation of the same stage,
which tests/test_host_core.py pins to the oracle.  Inputs are built to be hard for the run decomposition: tandem copies
of genes (long runs of anchors in adjacent blocks, many equal chain scores), low-complexity proteins, empty queries."""
import ctypes as C
import numpy as np
import pytest
import miniprot_amd as mpa
import golden  # noqa: F401  (puts tools/ on sys.path)
import gen_synth
import refbind

pytestmark = pytest.mark.gpu


def survivors(ctx, idx, mo, q, threads=4):
    L = mpa.lib()
    L.mpa_dbg_prechain_survivors.restype = C.c_int64
    L.mpa_dbg_prechain_survivors.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(mpa.MapOpt), C.POINTER(mpa.QBatch), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    off = np.zeros(len(q.seqs) + 1, np.int64)
    out = C.c_void_p()
    n = L.mpa_dbg_prechain_survivors(ctx.h if ctx else None, idx.h, C.byref(mo), C.byref(q.c), threads, off.ctypes.data, C.byref(out))
    assert n >= 0, mpa.last_error()
    a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
    L.mpa_free(out)
    return off, a


@pytest.mark.parametrize("mode", ["sift", "sift-small-segments"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_seeding_equals_host_seeding(seed, mode, monkeypatch):
    # k_seed_sift: per-query merge of the occurrence lists in block order + LDS sort + isolation filter, also with segments so
    # small that every query is cut into many of them
    if mode.endswith("segments"):
        monkeypatch.setenv("MPA_SIFT_SEG", "300")
    rng = np.random.default_rng(seed)
    contigs, prots, names, planted = gen_synth.generate(3000000, 2, 40, 20 + seed, return_planted=True)
    # tandem copies: re-plant the locus of a few genes several times back to back, on both strands of contig 0
    g = contigs[0]
    for k in range(6):
        src = int(rng.integers(0, len(g) - 20000))
        seg = g[src:src + int(rng.integers(2000, 9000))].copy()
        at = int(rng.integers(0, len(g) - 10 * len(seg)))
        for r in range(int(rng.integers(2, 6))):
            g[at + r * len(seg):at + (r + 1) * len(seg)] = seg
    idx = mpa.Index.from_nt4(contigs, ["chr1", "chr2"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 4))
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    extra = [b"M" + bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), 300).tolist()),       # unrelated
             b"MKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKKK", b"M", b"MA",             # low complexity, tiny
             bytes(planted[0]) + bytes(planted[1]) + bytes(planted[0])]                        # chimeric, repeated domain
    seqs = list(prots) + [bytes(p) for p in planted[:10]] + extra
    q = mpa.Queries(seqs)
    for flags in (0, 1):
        mo = mpa.default_mapopt()
        if flags:
            mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
            mo.max_occ = 200
        off_h, a_h = survivors(None, idx, mo, q)
        off_d, a_d = survivors(ctx, idx, mo, q)
        assert np.array_equal(off_h, off_d)
        assert np.array_equal(a_h, a_d)
        assert off_h[-1] > 0
    idx.close()
    ctx.close()


def test_sift_across_segment_boundaries(monkeypatch):
    """k_seed_sift cuts a query's block space into segments swept side by side: a 150 Mbp genome with a gene planted right across
    a segment boundary, so that kept anchors need the neighbouring segment's edge blocks and the rank base of the lower segments
    (host side of this test checked on CPU: query 0 keeps anchors in the three blocks below and the two blocks above the
    boundary)."""
    codon = {"A": "GCT", "C": "TGT", "D": "GAT", "E": "GAA", "F": "TTT", "G": "GGT", "H": "CAT", "I": "ATT", "K": "AAA", "L": "CTG", "M": "ATG",
             "N": "AAT", "P": "CCT", "Q": "CAA", "R": "CGT", "S": "TCT", "T": "ACT", "V": "GTT", "W": "TGG", "Y": "TAT"}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    contigs, prots, names = gen_synth.generate(150000000, 1, 30, 77)
    g = contigs[0]
    bblock = 4 * 131008
    boundary = bblock * 256
    p = bytes(prots[0]).decode()                             # the coding sequence of protein 0, back-translated, across the boundary
    nt = np.array([code[c] for a in p for c in codon.get(a, "GCT")], np.uint8)
    g[boundary - 600:boundary - 600 + len(nt)] = nt
    idx = mpa.Index.from_nt4(contigs, ["chr1"])
    ctx = mpa.Context(0)
    assert idx.build_kmers(4, ctx) == "gpu"
    idx.to_device(ctx)
    q = mpa.Queries([bytes(p) for p in prots])
    mo = mpa.default_mapopt()
    mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
    off_h, a_h = survivors(None, idx, mo, q, threads=16)
    for mode, seg in (("sift", None), ("sift", "2000"), ("sift", "256")):
        if seg:
            monkeypatch.setenv("MPA_SIFT_SEG", seg)          # 2 / ~12 segments per query, swept side by side: boundaries all over the genome
        off_d, a_d = survivors(ctx, idx, mo, q, threads=16)
        assert np.array_equal(off_h, off_d), (mode, seg)
        assert np.array_equal(a_h, a_d), (mode, seg)
    # the planted copies are really there: anchors of query 0 in the blocks either side of the boundary
    blocks = (a_h[off_h[0]:off_h[1]] >> 32).astype(np.int64)
    assert ((blocks >= bblock - 4) & (blocks < bblock)).any() and ((blocks >= bblock) & (blocks < bblock + 6)).any()
    idx.close()
    ctx.close()


def main_chains(ctx, idx, mo, q, threads=4):
    L = mpa.lib()
    L.mpa_dbg_main_chains.restype = C.c_int64
    L.mpa_dbg_main_chains.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(mpa.MapOpt), C.POINTER(mpa.QBatch), C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p)]
    off_u, off_a = np.zeros(len(q.seqs) + 1, np.int64), np.zeros(len(q.seqs) + 1, np.int64)
    ou, oa = C.c_void_p(), C.c_void_p()
    n_back = L.mpa_dbg_main_chains(ctx.h if ctx else None, idx.h, C.byref(mo), C.byref(q.c), threads, off_u.ctypes.data, C.byref(ou), off_a.ctypes.data, C.byref(oa))
    assert n_back >= 0, mpa.last_error()
    u = np.ctypeslib.as_array(C.cast(ou, C.POINTER(C.c_uint64)), (max(int(off_u[-1]), 1),))[:off_u[-1]].copy()
    a = np.ctypeslib.as_array(C.cast(oa, C.POINTER(C.c_uint64)), (max(int(off_a[-1]), 1),))[:off_a[-1]].copy()
    L.mpa_free(ou), L.mpa_free(oa)
    return n_back, off_u, u, off_a, a


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_chains_equal_host_chains(seed, monkeypatch):
    """Both chaining rounds on the device (k_chain_extract = the host's own extraction source compiled for the GPU, k_chain_fwd)
    against the host stages that tests/test_host_core.py pins to the oracle: the main chains (score, count, anchors, in the order
    mp_chain() returns them) of every query -- tandem copies of genes (many equal chain scores: the tie order of the reference's
    unstable sort decides which chain claims a shared anchor), low-complexity and tiny queries (problems of <= 64 anchors take the
    full-list path), default and -I style parameters"""
    rng = np.random.default_rng(seed)
    contigs, prots, names, planted = gen_synth.generate(3000000, 2, 40, 20 + seed, return_planted=True)
    g = contigs[0]
    for k in range(6):
        src = int(rng.integers(0, len(g) - 20000))
        seg = g[src:src + int(rng.integers(2000, 9000))].copy()
        at = int(rng.integers(0, len(g) - 10 * len(seg)))
        for r in range(int(rng.integers(2, 6))):
            g[at + r * len(seg):at + (r + 1) * len(seg)] = seg
    idx = mpa.Index.from_nt4(contigs, ["chr1", "chr2"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 4))
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    extra = [b"M" + bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), 300).tolist()), b"MLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLL", b"M", b"MA",
             bytes(planted[0]) + bytes(planted[1]) + bytes(planted[0])]
    q = mpa.Queries(list(prots) + [bytes(p) for p in planted[:10]] + extra)
    n_chains = 0
    for flags in (0, 1):
        mo = mpa.default_mapopt()
        if flags:
            mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
            mo.max_occ = 200
        _, hu_off, hu, ha_off, ha = main_chains(None, idx, mo, q)
        n_back, du_off, du, da_off, da = main_chains(ctx, idx, mo, q)
        assert np.array_equal(hu_off, du_off) and np.array_equal(hu, du), flags
        assert np.array_equal(ha_off, da_off) and np.array_equal(ha, da), flags
        assert n_back < len(q.seqs) // 2                           # the device really did most of them
        n_chains += int(hu_off[-1])
    assert n_chains > 50
    idx.close()
    ctx.close()


def test_sift_declines_a_block_that_cannot_fit_and_the_host_takes_over(monkeypatch):
    """k_seed_sift holds the anchors of one range in a 4 096-entry LDS buffer; a query whose seeds put more than that into ONE
    block (here: 5 000 identical k-mers of a poly-L protein over CTG repeats in the genome) is flagged by the kernel and
    seeded on the host, the other queries of the batch stay on the device -- same survivors as the host stage for all of them"""
    contigs, prots, names, planted = gen_synth.generate(2000000, 1, 20, 31, return_planted=True)
    g = contigs[0]
    for at in (300000, 300700, 900000, 1500000):
        g[at:at + 420] = np.tile(np.array([1, 3, 2], np.uint8), 140)   # CTG CTG ...: leucine codons, an ORF of 140 codons on the forward strand
    idx = mpa.Index.from_nt4(contigs, ["chr1"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 4))
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    # (the third odd query: 300 seeds in the same few blocks of an otherwise wide range -- one bucket of the in-LDS sort holds
    # hundreds of keys, which sends that range to the bitonic network)
    seqs = [bytes(p) for p in planted[:8]] + [b"M" + b"L" * 5000, b"L" * 1500 + bytes(planted[0]), bytes(planted[1]) + b"L" * 300] + [bytes(p) for p in planted[8:12]]
    q = mpa.Queries(seqs)
    mo = mpa.default_mapopt()
    mo.max_occ = 100000
    off_h, a_h = survivors(None, idx, mo, q)
    off_d, a_d = survivors(ctx, idx, mo, q)
    assert np.array_equal(off_h, off_d) and np.array_equal(a_h, a_d)
    assert off_h[9] - off_h[8] > 1000                            # the poly-L query does have (many) anchors in the repeats
    idx.close()
    ctx.close()


def test_device_refinement_scan_equals_host_scan():
    """k_refine_scan against the host's window scan: windows of awkward lengths (chunk size 2048 +-1, shorter than a k-mer),
    at contig ends, on both strands, over sequence rich in N and in stop codons, and a query that is made of the
    windows' own translations (many hits) -- the same (hash, position) sets"""
    rng = np.random.default_rng(5)
    L = mpa.lib()
    L.mpa_dbg_refine_hits.restype = C.c_int64
    L.mpa_dbg_refine_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    # a contig without stop codons in frame 0 for long stretches (long ORFs), N islands, and random sequence
    codons = [c for c in range(64) if c not in (48, 50, 56)]             # all but TAA TAG TGA
    orf = np.array([[c >> 4, c >> 2 & 3, c & 3] for c in rng.choice(codons, 6000)], dtype=np.uint8).reshape(-1)
    g = np.concatenate([rng.integers(0, 4, 3000).astype(np.uint8), orf, rng.integers(0, 4, 9000).astype(np.uint8), orf[::-1].copy() ^ 3, rng.integers(0, 4, 500).astype(np.uint8)])
    for at in rng.integers(0, len(g) - 40, 25):
        g[at:at + int(rng.integers(1, 30))] = 4
    idx = mpa.Index.from_nt4([g, rng.integers(0, 4, 700).astype(np.uint8)], ["c1", "c2"])
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    # the query: translation of stretches of the contig, so that many window k-mers are in its k-mer set
    aa_of = {}
    names, seqs = [], []
    import refbind
    nt_fwd = idx.get_nt(0, 3000, 3000 + 9000)
    codon_tab = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"
    prot = "".join("X" if max(nt_fwd[i:i + 3]) > 3 else codon_tab[int(nt_fwd[i]) << 4 | int(nt_fwd[i + 1]) << 2 | int(nt_fwd[i + 2])]
                   for i in range(0, 2400, 3)).replace("*", "X").encode()
    wins = []
    clen = [len(g), 700]
    for ln in (0, 5, 14, 15, 16, 89, 90, 2047, 2048, 2049, 4096, 6000, 20000):
        for vid in (0, 1):
            st = int(rng.integers(0, max(clen[0] - ln, 1)))
            wins.append((vid, st, min(ln, clen[0] - st)))
    wins += [(0, 0, clen[0]), (1, 0, clen[0]), (0, clen[0] - 3000, 3000), (1, clen[0] - 2100, 2100), (2, 0, 700), (3, 100, 600)]
    vid = np.array([w[0] for w in wins], np.int32)
    as_ = np.array([w[1] for w in wins], np.int64)
    ln_ = np.array([w[2] for w in wins], np.int32)

    def run(c):
        first = np.zeros(len(wins) + 1, np.int64)
        out = C.c_void_p()
        n = L.mpa_dbg_refine_hits(c.h if c else None, idx.h, 5, prot, len(prot), len(wins), vid.ctypes.data, as_.ctypes.data, ln_.ctypes.data, first.ctypes.data, C.byref(out))
        assert n >= 0, mpa.last_error()
        a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
        L.mpa_free(out)
        return first, a

    fh, ah = run(None)
    fd, ad = run(ctx)
    assert np.array_equal(fh, fd) and np.array_equal(ah, ad)
    assert fh[-1] > 500                                                   # the windows over the ORF stretch do hit
    idx.close()
    ctx.close()


@pytest.mark.parametrize("serial_run", ["48", "4"])
def test_device_chain_forward_equals_host_forward(serial_run, monkeypatch):
    """k_chain_fwd (the forward pass of mp_chain for any chaining problem, one thread per independent run of anchors) against
    the host pass that tests/test_host_core.py pins to the oracle: block anchors (main chain: long windows, planted chains,
    dense repeats that exercise max_skip and max_iter) and base-resolution anchors (refinement chains, the exact-gap branch
    of comput_sc), many problems per launch, empty and one-anchor problems in between"""
    # serial_run: runs longer than this are walked by a wavefront (k_chain_fwd_wave: 64 candidates at a time, the max_skip counter
    # replayed over ballots) instead of one thread; 4 sends almost every run there
    monkeypatch.setenv("MPA_CHAIN_SERIAL_RUN", serial_run)
    from test_host_core import _anchors, _long_chains
    L = mpa.lib()
    L.mpa_dbg_chain_forward.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [C.c_float] + [C.c_int32] * 4 + [C.c_void_p] * 4
    rng = np.random.default_rng(17)
    ctx = mpa.Context(0)

    def both(args, probs):
        first = np.zeros(len(probs) + 1, np.int64)
        np.cumsum([len(p) for p in probs], out=first[1:])
        a = np.ascontiguousarray(np.concatenate(probs + [np.zeros(0, np.uint64)]), dtype=np.uint64)
        res = []
        for c in (None, ctx.h):
            f, pr = np.full(len(a) + 1, -7, np.int32), np.full(len(a) + 1, -7, np.int32)
            rc = L.mpa_dbg_chain_forward(c, *args, len(probs), first.ctypes.data, a.ctypes.data, f.ctypes.data, pr.ctypes.data)
            assert rc == 0, mpa.last_error()
            res.append((f[:len(a)], pr[:len(a)]))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), (args, [len(p) for p in probs][:8])
        return int((res[0][1] >= 0).sum())

    linked = 0
    for it in range(12):                                        # main-chain shape: max_dist_x = max_intron, bbit 8
        probs = []
        for k in range(int(rng.choice([1, 7, 60]))):
            kind = int(rng.integers(0, 5))
            if kind == 0:
                probs.append(_anchors(rng, int(rng.choice([70, 500, 9000])), 3000000, 400, int(rng.choice([0, 3, 40]))))
            elif kind == 1:
                probs.append(_anchors(rng, int(rng.choice([65, 300, 3000])), int(rng.choice([50, 400, 3000])), 300, 20))       # dense: max_skip matters
            elif kind == 2:
                probs.append(np.unique(np.concatenate([_anchors(rng, 2000, 3000000, 900, 10), _long_chains(rng, 3000000, 3, "any")])))
            elif kind == 3:
                probs.append(np.zeros(0, np.uint64))
            else:
                probs.append(_anchors(rng, 1, 1000, 100, 0)[:1])
        args = [int(rng.choice([200000, 20000])), 1000, int(rng.choice([200000, 20000])), int(rng.choice([25, 2])), int(rng.choice([1000000, 50])), 0.75, int(rng.integers(0, 2)), 6, 8]
        linked += both(args, probs)
    for it in range(8):                                         # refinement shape: base resolution, kmer 5
        probs = []
        for k in range(int(rng.choice([1, 30]))):
            n = int(rng.choice([0, 1, 5, 80, 600]))
            dq = np.cumsum(rng.integers(1, 9, n))
            x = 1000 + dq * 3 + (rng.choice([0, 0, 1, -1, 300, 5000], n) * (rng.random(n) < 0.1)).cumsum()
            probs.append(np.unique((x.astype(np.uint64) << np.uint64(32)) | (20 + dq).astype(np.uint64)))
        args = [200000, 1000, 200000, int(rng.choice([25, 2])), 1000000, 0.75, int(rng.integers(0, 2)), 5, 0]
        linked += both(args, probs)
    assert linked > 5000
    ctx.close()


def raw_anchors(idx, mo, q, threads=4):
    L = mpa.lib()
    L.mpa_dbg_anchors.restype = C.c_int64
    L.mpa_dbg_anchors.argtypes = [C.c_void_p, C.POINTER(mpa.MapOpt), C.POINTER(mpa.QBatch), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    off = np.zeros(len(q.seqs) + 1, np.int64)
    out = C.c_void_p()
    n = L.mpa_dbg_anchors(idx.h, C.byref(mo), C.byref(q.c), threads, off.ctypes.data, C.byref(out))
    assert n >= 0, mpa.last_error()
    a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
    L.mpa_free(out)
    return off, a


@pytest.mark.skipif(not refbind.have_ref(), reason="oracle/_ref/libminiprot_ref.so not built")
@pytest.mark.parametrize("seed", [5, 6])
def test_device_chaining_rounds_equal_the_reference_mp_chain(seed):
    """Both chaining rounds of the device path against the REFERENCE ITSELF (oracle/_ref/libminiprot_ref.so, mp_chain, chain.c:160-219),
    not against the host stage: the anchors of every query (map.c:163-178) go through the reference's pre-chain call (map.c:188:
    min_cnt 2, min_sc 0, one block of reach) and its sort (map.c:191-192) -- which must give k_seed_sift + k_prechain_fwd +
    k_chain_extract(set)'s survivors -- and those through the reference's main-chain call (map.c:194-195), which must give
    k_chain_fwd / k_chain_fwd_wave + k_chain_extract's chains: u[] (score << 32 | anchors) and the anchors, chain by chain, in the
    order mp_chain() returns them.  Tandem gene copies: many equal chain scores, i.e. the tie order of the reference's unstable
    radix sort decides which chain claims a shared anchor."""
    rng = np.random.default_rng(seed)
    contigs, prots, names, planted = gen_synth.generate(3000000, 2, 40, 30 + seed, return_planted=True)
    g = contigs[0]
    for k in range(6):
        src = int(rng.integers(0, len(g) - 20000))
        seg = g[src:src + int(rng.integers(2000, 9000))].copy()
        at = int(rng.integers(0, len(g) - 10 * len(seg)))
        for r in range(int(rng.integers(2, 6))):
            g[at + r * len(seg):at + (r + 1) * len(seg)] = seg
    idx = mpa.Index.from_nt4(contigs, ["chr1", "chr2"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 4))
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    extra = [b"M" + bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), 300).tolist()), b"MA", bytes(planted[0]) + bytes(planted[1]) + bytes(planted[0])]
    q = mpa.Queries(list(prots) + [bytes(p) for p in planted[:10]] + extra)
    kmer, bbit = 6, 8                                            # mp_idxopt_init (options.c:14-22): what Index.from_nt4 builds with
    n_chains = n_surv = 0
    for flags in (0, 1):
        mo = mpa.default_mapopt()
        if flags:
            mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
            mo.max_occ = 200
        is_spliced = 0 if (mo.flag & 0x1) else 1                  # MPA_MF_NO_SPLICE
        a_off, a_all = raw_anchors(idx, mo, q)
        s_off, s_dev = survivors(ctx, idx, mo, q)
        n_back, du_off, du, da_off, da = main_chains(ctx, idx, mo, q)
        w = 1 << bbit
        for i in range(len(q.seqs)):
            a = a_all[a_off[i]:a_off[i + 1]]
            _, pre = refbind.ref_chain(a, (w, w, w, mo.max_chn_max_skip, mo.max_chn_iter, 2, 0, mo.chn_coef_log, is_spliced, kmer, bbit))
            pre = np.sort(pre)                                     # radix_sort_mp64, map.c:191
            assert np.array_equal(pre, s_dev[s_off[i]:s_off[i + 1]]), ("pre-chain survivors", flags, i)
            u, ca = refbind.ref_chain(pre, (mo.max_intron, mo.max_gap, mo.bw, mo.max_chn_max_skip, mo.max_chn_iter, mo.min_chn_cnt, mo.min_chn_sc, mo.chn_coef_log,
                                            is_spliced, kmer, bbit))
            assert np.array_equal(u, du[du_off[i]:du_off[i + 1]]), ("main chains: u", flags, i)
            assert np.array_equal(ca, da[da_off[i]:da_off[i + 1]]), ("main chains: anchors", flags, i)
            n_chains += len(u)
            n_surv += len(pre)
    assert n_chains > 50 and n_surv > 1000
    idx.close()
    ctx.close()
