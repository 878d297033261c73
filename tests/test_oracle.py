"""The oracle (oracle/*.c) is pinned two ways:
  * against the committed golden vectors produced by the real reference (always), and
  * against the compiled reference itself (oracle/_ref) by live fuzzing, when it is present."""
import ctypes as C
import numpy as np
import pytest
import refbind
import golden
from dpgen import make_task, make_ss


def test_dp_oracle_matches_golden_vectors(oracle_built):
    z = np.load(golden.path("dp_vectors.npz"))
    P0 = z["mat"]
    nt_off = np.concatenate([[0], np.cumsum(z["nt_len"])])
    aa_off = np.concatenate([[0], np.cumsum(z["aa_len"])])
    cg_off = np.concatenate([[0], np.cumsum(z["cig_len"])])
    n = len(z["flag"])
    assert n == 900
    for k in range(n):
        nt = bytes(z["nt"][nt_off[k]:nt_off[k + 1]])
        aa = bytes(z["aa"][aa_off[k]:aa_off[k + 1]])
        P = refbind.DpParams(P0, io=int(z["io"][k]))
        got = refbind.ora_nasw(nt, aa, P, int(z["flag"][k]))
        want = (int(z["res"][k][0]), int(z["res"][k][1]), int(z["res"][k][2]), [int(x) for x in z["cig"][cg_off[k]:cg_off[k + 1]]])
        assert got == want, "vector %d flag %d" % (k, z["flag"][k])


needs_ref = pytest.mark.skipif(not refbind.have_ref(), reason="compiled reference (oracle/_ref) not present")


@needs_ref
def test_dp_oracle_vs_reference_fuzz(oracle_built):
    rng = np.random.default_rng(101)
    for it in range(250):
        fs = int(rng.choice([23, 17, 10]))
        P = refbind.DpParams(refbind.mapping_matrix(fs), go=int(rng.choice([11, 5])), ge=int(rng.choice([1, 2])),
                             io=int(rng.choice([29, 19, 40])), fs=fs, xdrop=int(rng.choice([100, 20, 1000])),
                             end_bonus=int(rng.choice([5, 0])), sp=[(8, 15, 21, 30, 0, 0), (8, 15, 21, 30, 4, 4), (0, 0, 0, 0, 0, 0)][rng.integers(0, 3)],
                             ie_coef=float(rng.choice([0.5, 0.0, 1.0])))
        nt, aa = make_task(rng, max_intron=int(rng.choice([100, 400, 2000])), flank=int(rng.choice([10, 60, 1000])))
        ss = make_ss(rng, len(nt)) if rng.random() < 0.2 else None
        for flag in (1, 2, 4, 0):
            assert refbind.ref_nasw(nt, aa, P, flag, ss) == refbind.ora_nasw(nt, aa, P, flag, ss), (it, flag)


@needs_ref
def test_chain_sort_sketch_oracle_vs_reference(oracle_built):
    r, o = refbind.ref(), refbind.ora()
    libc = refbind._libc
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    rng = np.random.default_rng(3)

    def ref_chain(a, args):
        n = len(a)
        p = libc.malloc(max(8 * n, 8))
        C.memmove(p, a.ctypes.data, 8 * n)
        n_u, u = C.c_int32(0), C.c_void_p(0)
        res = r.mp_chain(*args, n, p, C.byref(n_u), C.byref(u), None)
        if n_u.value == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
        uu = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), (n_u.value,)).copy()
        aa = np.ctypeslib.as_array(C.cast(res, C.POINTER(C.c_uint64)), (int((uu & 0xffffffff).sum()),)).copy()
        libc.free(res), libc.free(u)
        return uu, aa

    def ora_chain(a, args):
        a = a.copy()
        u = np.zeros(len(a) + 1, np.uint64)
        no = C.c_int64(0)
        nu = o.mpo_chain(*args, len(a), a.ctypes.data, u.ctypes.data, C.byref(no))
        return u[:nu].copy(), a[:no.value].copy()

    for it in range(400):
        bbit = int(rng.choice([0, 8]))
        n = int(rng.choice([0, 1, 5, 50, 300, 2000]))
        if n:
            xs, ys = [], []
            nc = max(1, n // 20)
            for c in range(nc):
                x0, y0, m = rng.integers(0, 200000 if bbit == 0 else 3000), rng.integers(0, 500), n // nc
                dy = np.cumsum(rng.integers(1, 12, m))
                dx = dy * 3 + (rng.choice([0, 0, 0, 1, -1, 300, 5000, 90000], m) * (rng.random(m) < 0.1)).cumsum() if bbit == 0 else (dy * 3) // 256 + (rng.random(m) < 0.1) * rng.integers(0, 300, m)
                xs.append(x0 + dx), ys.append(y0 + dy)
            a = np.unique((np.concatenate(xs).astype(np.uint64) << np.uint64(32)) | (np.concatenate(ys).astype(np.uint64) & np.uint64(0x7fffffff)))
        else:
            a = np.zeros(0, np.uint64)
        if bbit:
            args = [256, 256, 256, 25, 1000000, 2, 0, 0.75, 1, 6, 8] if rng.random() < 0.5 else [200000, 1000, 200000, 25, int(rng.choice([1000000, 50])), 3, 0, 0.75, int(rng.integers(0, 2)), 6, 8]
        else:
            args = [200000, 1000, 200000, int(rng.choice([25, 2])), 1000000, 3, 0, 0.75, int(rng.integers(0, 2)), 5, 0]
        ua, aa = ref_chain(a, args)
        ub, ab = ora_chain(a, args)
        assert np.array_equal(ua, ub) and np.array_equal(aa, ab), (it, n, args)

    r.radix_sort_mp64.argtypes = [C.c_void_p, C.c_void_p]
    r.radix_sort_mp128x.argtypes = [C.c_void_p, C.c_void_p]
    for it in range(60):
        n = int(rng.choice([3, 64, 65, 200, 5000, 50000]))
        z = np.zeros((n, 2), np.uint64)
        z[:, 0] = rng.integers(0, int(rng.choice([4, 300, 1 << 30])), n)
        z[:, 1] = np.arange(n)
        b, c = z.copy(), z.copy()
        r.radix_sort_mp128x(b.ctypes.data, b.ctypes.data + 16 * n)
        o.mpo_radix_sort128x(c.ctypes.data, c.ctypes.data + 16 * n)
        assert np.array_equal(b, c)
