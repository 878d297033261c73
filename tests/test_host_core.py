"""Host stages with reference-defined tie behaviour (chaining + its chain-end sort, the anchor sort), fuzzed through the
library's test hooks against the oracle (which tests/test_oracle.py pins to the reference).  The product's chain
extraction replays the reference's in-place MSD radix sort sparsely (host_core.cpp: sorted_chain_ends_sparse), so the
inputs here are built to have many equal chain scores."""
import ctypes as C
import numpy as np
import pytest
import miniprot_amd as mpa
import refbind

CHAIN_ARGS = [C.c_int32] * 7 + [C.c_float] + [C.c_int32] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]


def _chain(fn, a, args):
    a = a.copy()
    u = np.zeros(len(a) + 1, np.uint64)
    no = C.c_int64(0)
    nu = fn(*args, len(a), a.ctypes.data, u.ctypes.data, C.byref(no))
    return u[:nu].copy(), a[:no.value].copy()


def _anchors(rng, n, n_block, qlen, n_planted):
    """random background + planted collinear runs + duplicated fragments (equal scores on purpose)"""
    blk = rng.integers(0, n_block, n)
    q = rng.integers(5, qlen, n)
    xs, ys = [blk], [q]
    for _ in range(n_planted):
        b0, q0, m = int(rng.integers(0, n_block)), int(rng.integers(5, qlen // 2)), int(rng.integers(2, 12))
        dq = np.cumsum(rng.integers(1, 9, m))
        for rep in range(int(rng.integers(1, 4))):           # the same fragment again a few blocks away / shifted by one residue
            xs.append(b0 + (dq * 3) // 256 + rep * int(rng.integers(0, 3)))
            ys.append(q0 + dq + (rep if rng.random() < 0.5 else 0))
    a = (np.concatenate(xs).astype(np.uint64) << np.uint64(32)) | (np.concatenate(ys).astype(np.uint64) & np.uint64(0x7fffffff))
    return np.unique(a)


def _long_chains(rng, n_block, n_copies, where):
    """the same long collinear fragment (chain score well above 255) planted several times: equal high scores, placed at
    the start / end / anywhere of the block range so that they fall inside and behind the first bucket's region"""
    m = int(rng.choice([50, 90, 160]))
    dq = np.cumsum(rng.integers(1, 4, m))
    out = []
    for c in range(n_copies):
        b0 = {"start": 3 + 40 * c, "end": n_block - 400 + 40 * c, "any": int(rng.integers(0, n_block - 100))}[where]
        jitter = 0 if rng.random() < 0.6 else int(rng.integers(0, 2))
        x = (b0 + (dq * 3) // 256).astype(np.uint64)
        out.append((x << np.uint64(32)) | (10 + jitter + dq).astype(np.uint64))
    return np.concatenate(out)


def test_chain_anchors_matches_oracle(oracle_built):
    o, L = refbind.ora(), mpa.lib()
    L.mpa_dbg_chain.restype = C.c_int32
    L.mpa_dbg_chain.argtypes = CHAIN_ARGS
    o.mpo_chain.restype = C.c_int32
    o.mpo_chain.argtypes = CHAIN_ARGS
    rng = np.random.default_rng(11)
    pre = [256, 256, 256, 25, 1000000, 2, 0, 0.75, 1, 6, 8]
    n_sparse = 0
    for it in range(360):
        kind = it % 5
        if kind == 0:     # sparse: almost every anchor is alone (the large-genome regime), a few chains
            a = _anchors(rng, int(rng.choice([70, 500, 5000, 60000])), 3000000, 400, int(rng.choice([0, 3, 40])))
        elif kind == 1:   # dense: everything chains with everything, many ties
            a = _anchors(rng, int(rng.choice([65, 300, 3000])), int(rng.choice([50, 400, 3000])), 300, 20)
        elif kind == 2:   # tail-heavy: chained anchors concentrated at the end / start of the array
            a = _anchors(rng, 2000, 3000000, 400, 0)
            hot = _anchors(rng, 300, 40, 300, 30) + (np.uint64(0 if rng.random() < 0.5 else 2999000) << np.uint64(32))
            a = np.unique(np.concatenate([a, hot]))
        elif kind == 4:   # equal chain scores above 255 (two digit levels of the chain-end sort), few or many other anchors
            bg = _anchors(rng, int(rng.choice([0, 40, 3000, 40000])), 3000000, 900, int(rng.choice([0, 30])))
            a = np.unique(np.concatenate([bg, _long_chains(rng, 3000000, int(rng.integers(1, 5)), str(rng.choice(["start", "end", "any"])))]))
        else:
            a = _anchors(rng, int(rng.choice([0, 1, 30, 64, 65, 66])), 2000, 200, 3)
        args = list(pre)
        if it % 7 == 3:
            args = [200000, 1000, 200000, 25, int(rng.choice([1000000, 50])), 3, 0, 0.75, int(rng.integers(0, 2)), 6, 8]
        if it % 11 == 5:
            args[5], args[6] = int(rng.choice([1, 2, 4])), int(rng.choice([0, 6, 7, 20]))      # min_cnt / min_sc off the fast paths
        ua, aa = _chain(L.mpa_dbg_chain, a, args)
        ub, ab = _chain(o.mpo_chain, a, args)
        assert np.array_equal(ua, ub) and np.array_equal(aa, ab), (it, len(a), args)
        n_sparse += len(a) > 64 and args[5] > 1 and args[6] <= args[9]
    assert n_sparse > 100
    # base-resolution chaining (refinement), kmer 5
    for it in range(60):
        n = int(rng.choice([5, 80, 600]))
        dq = np.cumsum(rng.integers(1, 9, n))
        x = 1000 + dq * 3 + (rng.choice([0, 0, 1, -1, 300, 5000], n) * (rng.random(n) < 0.1)).cumsum()
        a = np.unique((x.astype(np.uint64) << np.uint64(32)) | (20 + dq).astype(np.uint64))
        args = [200000, 1000, 200000, int(rng.choice([25, 2])), 1000000, 3, 0, 0.75, int(rng.integers(0, 2)), 5, 0]
        ua, aa = _chain(L.mpa_dbg_chain, a, args)
        ub, ab = _chain(o.mpo_chain, a, args)
        assert np.array_equal(ua, ub) and np.array_equal(aa, ab), ("refine", it, n, args)


def test_prechain_through_sparse_view_matches_oracle(oracle_built):
    """the GPU seeding stage hands chain_extract() only the anchors that have a neighbour (a sparse ChainView): same chains"""
    o, L = refbind.ora(), mpa.lib()
    o.mpo_chain.restype = C.c_int32
    o.mpo_chain.argtypes = CHAIN_ARGS
    L.mpa_dbg_prechain_sparse.restype = C.c_int32
    L.mpa_dbg_prechain_sparse.argtypes = [C.c_int32] * 3 + [C.c_float] + [C.c_int32] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    rng = np.random.default_rng(12)
    big_f = 0
    for it in range(360):
        kind = it % 6
        if kind == 0:
            a = _anchors(rng, int(rng.choice([70, 500, 5000, 60000])), 3000000, 400, int(rng.choice([0, 3, 40])))
        elif kind == 1:
            a = _anchors(rng, int(rng.choice([65, 300, 3000])), int(rng.choice([50, 400, 3000])), 300, 20)
        elif kind == 2:
            a = _anchors(rng, 2000, 3000000, 400, 0)
            hot = _anchors(rng, 300, 40, 300, 30) + (np.uint64(0 if rng.random() < 0.5 else 2999000) << np.uint64(32))
            a = np.unique(np.concatenate([a, hot]))
        elif kind == 3:   # a long collinear chain: chain scores above 255 (the dense fallback with absent roots)
            m = int(rng.choice([60, 150]))
            dq = np.cumsum(rng.integers(1, 4, m))
            x = (5000 + (dq * 3) // 256).astype(np.uint64)
            a = np.unique(np.concatenate([_anchors(rng, 3000, 3000000, 900, 5), (x << np.uint64(32)) | (10 + dq).astype(np.uint64)]))
        elif kind == 5:
            bg = _anchors(rng, int(rng.choice([0, 40, 3000, 40000])), 3000000, 900, int(rng.choice([0, 30])))
            a = np.unique(np.concatenate([bg, _long_chains(rng, 3000000, int(rng.integers(1, 5)), str(rng.choice(["start", "end", "any"])))]))
        else:
            a = _anchors(rng, int(rng.choice([0, 1, 30, 64, 65, 66])), 2000, 200, 3)
        max_iter = int(rng.choice([1000000, 1000000, 20]))
        ua, aa = _chain(o.mpo_chain, a, [256, 256, 256, 25, max_iter, 2, 0, 0.75, 1, 6, 8])
        b = a.copy()
        u = np.zeros(len(a) + 1, np.uint64)
        no = C.c_int64(0)
        nu = L.mpa_dbg_prechain_sparse(256, 25, max_iter, 0.75, 1, 6, 8, len(b), b.ctypes.data, u.ctypes.data, C.byref(no))
        assert np.array_equal(ua, u[:nu]) and np.array_equal(aa, b[:no.value]), (it, len(a))
        big_f += len(ua) > 0 and int((ua >> np.uint64(32)).max()) > 255
    assert big_f > 10


def test_pair_sort_tie_order_matches_oracle(oracle_built):
    o, L = refbind.ora(), mpa.lib()
    L.mpa_dbg_sort_pairs.argtypes = [C.c_void_p, C.c_void_p]
    L.mpa_dbg_sort_pairs.restype = None
    rng = np.random.default_rng(5)
    for it in range(80):
        n = int(rng.choice([3, 64, 65, 200, 5000, 50000]))
        z = np.zeros((n, 2), np.uint64)
        z[:, 0] = rng.integers(0, int(rng.choice([4, 40, 300, 70000, 1 << 30])), n) + int(rng.choice([0, 6, 1 << 20]))
        z[:, 1] = np.arange(n)
        b, c = z.copy(), z.copy()
        L.mpa_dbg_sort_pairs(b.ctypes.data, b.ctypes.data + 16 * n)
        o.mpo_radix_sort128x(c.ctypes.data, c.ctypes.data + 16 * n)
        assert np.array_equal(b, c), (it, n)


def test_anchor_sort_is_a_full_key_sort():
    L = mpa.lib()
    L.mpa_dbg_sort_anchors.argtypes = [C.c_void_p, C.c_int64]
    L.mpa_dbg_sort_anchors.restype = None
    rng = np.random.default_rng(6)
    for it in range(40):
        n = int(rng.choice([0, 1, 200, 257, 5000, 200000]))
        n_seed = int(rng.choice([1, 7, 300]))
        qpos = np.sort(rng.choice(2000, n_seed, replace=False))
        # seed by seed in ascending query position, blocks in any order within a seed
        q = np.sort(rng.choice(qpos, n))
        blk = rng.integers(0, int(rng.choice([5, 9000, 1 << 14, 1 << 27, (1 << 32) - 1])), n).astype(np.uint64)
        a = (blk << np.uint64(32)) | q.astype(np.uint64)
        b = a.copy()
        L.mpa_dbg_sort_anchors(b.ctypes.data, len(b))
        assert np.array_equal(b, np.sort(a)), (it, n)
