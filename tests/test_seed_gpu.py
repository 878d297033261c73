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


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_seeding_equals_host_seeding(seed):
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
