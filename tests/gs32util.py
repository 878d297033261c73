"""Helpers of the 32-bit operator's tests: the model call (host instance of gs32_core.h inside libmpamd.so) and case generators."""
import ctypes as C
import numpy as np
import miniprot_amd as mpa
import refbind
from dpgen import make_task, make_ss

GS32_NEG = -0x40000000


def model_gs32(nt, aa, P, flag, ss=None, want_tb=True):
    """(score, cigar list | None) of the host model, fed like the device kernel: records + int16 profile."""
    nl, al = len(nt), len(aa)
    rec, aas = refbind.ora_prep_rec(nt, aa, P, flag, ss)
    ncol = (al + 7) // 8 * 8
    prof = np.full((22, ncol), -32768, dtype=np.int16)
    prof[:, :al] = P.mat.astype(np.int16)[:, aas]
    rec = np.ascontiguousarray(np.concatenate([rec, np.zeros(4, np.uint32)]))
    tb = np.zeros((max(nl, 1), ncol), dtype=np.uint16) if want_tb else None
    L = mpa.lib()
    L.mpa_dbg_gs32_model.restype = C.c_int32
    L.mpa_dbg_gs32_model.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    sc = L.mpa_dbg_gs32_model(nl, al, rec.ctypes.data, prof.ctypes.data, ncol, P.go, P.ge, P.io, P.fs, tb.ctypes.data if want_tb else None, ncol)
    cig = refbind.ora_walk_back(tb, nl, al, ncol) if want_tb else None
    return sc, cig


def gs32_cases(rng, n):
    """(nt, aa, P, flag, ss) with ordinary scores, scores far outside int16 (matrix x 60: gs16b would saturate), other
    penalties, splice tracks, left-extension preparation, tiny windows."""
    base = refbind.mapping_matrix(23)
    out = []
    for k in range(n):
        r = rng.random()
        fs = int(rng.choice([23, 17]))
        mat = refbind.mapping_matrix(fs)
        scale = 1
        if r < 0.3:
            scale = int(rng.choice([11, 40]))                       # x 11: |mat| <= 127 still int8; x 40 only for the small entries
            mat = np.clip(mat.astype(np.int32) * scale, -128, 127).astype(np.int8)
        go = int(rng.choice([11, 5, 0, 300, 5000]))
        P = refbind.DpParams(mat, go=go, ge=int(rng.choice([1, 2, 9])), io=int(rng.choice([29, 19, 400, 9000])), fs=fs,
                             sp=[(8, 15, 21, 30, 0, 0), (8, 15, 21, 30, 4, 4), (0, 0, 0, 0, 0, 0)][rng.integers(0, 3)])
        u = rng.random()
        if u < 0.1:
            nt, aa = bytes(rng.integers(0, 5, int(rng.integers(0, 9))).astype(np.uint8)), bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), int(rng.integers(1, 6))).astype(np.uint8))
        elif u < 0.3:                                                   # long, nearly identical, scaled matrix: scores of 40 000 .. 150 000
            mat = np.clip(refbind.mapping_matrix(fs).astype(np.int32) * 11, -128, 127).astype(np.int8)
            P = refbind.DpParams(mat, go=P.go, ge=P.ge, io=P.io, fs=fs, sp=P.sp)
            nt, aa = make_task(rng, al=int(rng.integers(500, 1600)), p_sub=0.02, p_indel=0.004, p_intron=0.01, max_intron=300, flank=int(rng.choice([10, 200])))
        else:
            nt, aa = make_task(rng, al=int(rng.choice([rng.integers(1, 12), rng.integers(8, 80), rng.integers(60, 400)])),
                               max_intron=int(rng.choice([100, 400, 2000])), flank=int(rng.choice([10, 60, 600])))
        ss = make_ss(rng, len(nt)) if rng.random() < 0.2 and len(nt) > 0 else None
        flag = int(rng.choice([1, 1, 1, 0, 3, 4]))                      # CIGAR / score only / CIGAR + left flag (prep reversed, no traceback) / right flag
        if len(nt) < 3 and (flag & 2):
            flag = 1                                                    # (ns_prep_seq_left writes out of bounds for an empty window: not a case to compare)
        out.append((nt, aa, P, flag, ss))
    return out
