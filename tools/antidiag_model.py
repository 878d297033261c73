"""CPU model of dp_antidiag.hip (one call, one int16 half of one 32-lane group): the same step order, delay lines, prologue mask,
running row maximum, history and slow path as the kernel, in plain Python on the oracle's row records -- checked against the
oracle (python tools/antidiag_model.py [n_cases]).  Test infrastructure: what the kernel must do, debuggable without a GPU."""
import math
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import refbind
from dpgen import make_task

NEG = -32768


def sat(x):
    return NEG if x < NEG else 32767 if x > 32767 else x


def log2_poly(x):
    z = np.float32(x).view(np.uint32)
    r = np.float32(int((int(z) >> 23) & 255) - 128)
    z2 = (np.uint32(int(z) & ~(255 << 23)) + np.uint32(127 << 23)).view(np.float32) if False else np.uint32((int(z) & ~(255 << 23)) + (127 << 23)).view(np.float32)
    r = np.float32(r + np.float32(np.float32(np.float32(np.float32(-0.34484843) * z2) + np.float32(2.02466578)) * z2) - np.float32(0.67487759))
    return r


def pen_of(P, x):
    return 0 if x < 2 else int(np.float32(np.float32(P.ie_coef) * log2_poly(x)) + np.float32(.5))


def s8(b):
    return b - 256 if b >= 128 else b


def antidiag(rec, aas, al, P, G=32):
    nl = len(rec)
    ncol = (al + 7) // 8 * 8
    mat = np.asarray(P.mat, dtype=np.int64).reshape(22, 22)
    prof = [[int(mat[a][aas[c]]) if c < al else NEG for a in range(22)] for c in range(G)]
    def record(i):
        if 0 <= i < nl:
            r = int(rec[i])
            return (r & 0xff) // 2, s8(r >> 8 & 0xff), r >> 16 & 0xff, s8(r >> 24 & 0xff)
        return 21, 0, P.ge, 0                                   # (garbage rows: never reach a result)
    go, ge, io, fs, eb, xdrop = P.go, P.ge, P.io, P.fs, P.end_bonus, P.xdrop
    # per-lane state
    Hr = [[NEG] * 3 for _ in range(G)]; Hs = [[NEG] * 3 for _ in range(G)]; Dr = [[NEG] * 3 for _ in range(G)]
    dn = [[0] * 3 for _ in range(G)]; ac = [[0] * 3 for _ in range(G)]
    A = [NEG] * G; B = [NEG] * G; Cc = [NEG] * G
    Hs[0][2], Hs[0][1], Hs[0][0] = 0, -fs, -fs
    for j in range(G):                                          # as if steps 0 and 1 had run: the records of rows 0 - j and 1 - j
        _, dn[j][1], _, ac[j][1] = record(0 - j)
        _, dn[j][0], _, ac[j][0] = record(1 - j)
    Hlast = [NEG] * G; Ilast = [NEG] * G; Mlast = [NEG] * G
    hist = [[NEG] * G for _ in range(32)]
    Gs, maxi, maxsc, done = NEG, -1, None, False
    Hbest = [NEG] * G
    limit = nl + ncol - 1
    t = 2
    while t < limit and not done:
        K = (t - 2) % 3
        R1, R2, R3 = (3 - K) % 3, (4 - K) % 3, (5 - K) % 3
        newH = [0] * G; newI = [0] * G; newM = [0] * G
        for j in range(G):
            i = t - j
            aa, don, gei, acc = record(i)
            S = prof[j][aa]
            Lh = Hlast[j - 1] if j > 0 else NEG
            Li = Ilast[j - 1] if j > 0 else NEG
            Lm = Mlast[j - 1] if j > 0 else NEG
            dn[j][R3], ac[j][R3] = don, acc
            h = sat(Hs[j][R3] + S)
            x = sat(max(sat(Hr[j][R3] - go), Dr[j][R3]) - gei); Dr[j][R3] = x; h = max(h, x)
            u = sat(Hr[j][R1] - io); x = max(sat(u - dn[j][R2]), A[j]); A[j] = x; h = max(h, sat(x - ac[j][R3]))
            u = sat(Hs[j][R1] - io); x = max(sat(u - dn[j][R1]), B[j]); B[j] = x; h = max(h, sat(x - ac[j][R2]))
            x = max(sat(u - dn[j][R3]), Cc[j]); Cc[j] = x; h = max(h, sat(x - ac[j][R1]))
            x = max(Hr[j][R1], Hr[j][R2], Hs[j][R1], Hs[j][R2]); h = max(h, sat(x - fs))
            ii = sat(max(sat(Lh - go), Li) - ge)
            h = max(h, ii)
            if i < 2:
                h = NEG
            Hr[j][R3] = h; Hs[j][R3] = Lh
            newH[j], newI[j] = h, ii
            newM[j] = max(Lm, sat(h + (eb if j == al - 1 else 0)))
            hist[t & 31][j] = h
            if t == 2 and j == 0:
                Hs[j][R1] = NEG; Hs[j][R2] = NEG
        Hlast, Ilast, Mlast = newH, newI, newM
        ih = t - (ncol - 1)
        if 2 <= ih < nl:
            pen = pen_of(P, ih - 3 * al)
            m = sat(newM[ncol - 1] - pen)
            if m > Gs:
                Gs, maxi, maxsc = m, ih, m + pen
                for j in range(G):
                    back = ncol - 1 - j
                    Hbest[j] = hist[(t - back) & 31][j] if back >= 0 else NEG
            if Gs - m > xdrop:
                done = True
        t += 1
    if maxi < 0:
        return 0, 0, None
    first = None
    for j in range(al):
        v = Hbest[j] + (eb if j == al - 1 else 0)
        if v == maxsc:
            first = j
            break
    return maxi + 1, (first + 1) if first is not None else al + 1, maxsc


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(41)
    bad = tot = 0
    for (go, ge, io, fs, xd, eb, sp, coef) in [(11, 1, 29, 23, 100, 5, (8, 15, 21, 30, 0, 0), 0.5), (11, 1, 29, 23, 15, 5, (8, 15, 21, 30, 0, 0), 0.5),
                                                (5, 2, 40, 17, 50, 0, (8, 15, 21, 30, 4, 4), 1.0)]:
        P = refbind.DpParams(refbind.mapping_matrix(fs), go=go, ge=ge, io=io, fs=fs, xdrop=xd, end_bonus=eb, sp=sp, ie_coef=coef)
        for it in range(n):
            al = int(rng.integers(17, 33))
            nt, aa = make_task(rng, al=al, flank=int(rng.choice([1, 5, 40, 300])), p_indel=0.0 if it % 3 else 0.02)
            if not 17 <= len(aa) <= 32:
                continue
            for flag in (2, 4):
                rec, aas = refbind.ora_prep_rec(nt, aa, P, flag)
                got = antidiag(rec, aas, len(aa), P)
                want = refbind.ora_nasw(nt, aa, P, flag)[:3]
                tot += 1
                if tuple(got) != tuple(want):
                    bad += 1
                    print("case", it, "flag", flag, "nl", len(nt), "al", len(aa), "model", got, "oracle", want)
    print("%d mismatches in %d calls" % (bad, tot))
