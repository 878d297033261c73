"""ns_global_gs32 / ns_global_gs32b (nasw.h:129,132): the host instance of gs32_core.h -- the very source the device kernel
k_gs32 is compiled from -- against the reference's function in its -msse4.1 build (oracle/_ref/libnasw_sse41.so; gs32_core.h and
oracle/Makefile say why that build) and against golden vectors that build produced (tests/golden/gs32_vectors.npz)."""
import numpy as np
import pytest
import refbind
import golden
from gs32util import model_gs32, gs32_cases, GS32_NEG

needs_ref41 = pytest.mark.skipif(not refbind.have_ref41(), reason="oracle/_ref/libnasw_sse41.so not present")


def check(case, want):
    nt, aa, P, flag, ss = case
    want_tb = bool(flag & 1) and not (flag & 6)
    sc, cig = model_gs32(nt, aa, P, flag, ss, want_tb=want_tb)
    assert (len(nt), len(aa)) == want[:2]
    assert sc == want[2], (sc, want[2], len(nt), len(aa), flag)
    assert (cig or []) == want[3], (len(nt), len(aa), flag)


def test_model_matches_golden_vectors(oracle_built):
    z = np.load(golden.path("gs32_vectors.npz"))
    n = len(z["flag"])
    assert n >= 150
    nt_off, aa_off = np.concatenate([[0], np.cumsum(z["nt_len"])]), np.concatenate([[0], np.cumsum(z["aa_len"])])
    cg_off, ss_off = np.concatenate([[0], np.cumsum(z["cig_len"])]), np.concatenate([[0], np.cumsum(z["ss_len"])])
    big = 0
    for k in range(n):
        nt, aa = bytes(z["nt"][nt_off[k]:nt_off[k + 1]]), bytes(z["aa"][aa_off[k]:aa_off[k + 1]])
        ss = bytes(z["ss"][ss_off[k]:ss_off[k + 1]]) if z["ss_len"][k] else None
        go, ge, io, fs = (int(x) for x in z["par"][k][:4])
        P = refbind.DpParams(z["mat"][k], go=go, ge=ge, io=io, fs=fs, sp=tuple(int(x) for x in z["par"][k][4:10]))
        want = (len(nt), len(aa), int(z["score"][k]), [int(x) for x in z["cig"][cg_off[k]:cg_off[k + 1]]])
        check((nt, aa, P, int(z["flag"][k]), ss), want)
        big += want[2] > 32767
    assert big >= 10                                                     # the set really leaves the int16 range


@needs_ref41
def test_model_vs_reference_fuzz(oracle_built):
    rng = np.random.default_rng(4321)
    n_big = 0
    for case in gs32_cases(rng, 220):
        nt, aa, P, flag, ss = case
        want = refbind.ref_gs32(nt, aa, P, flag, ss)
        check(case, want)
        n_big += want[2] > 32767
    assert n_big >= 15


@needs_ref41
def test_reference_builds_agree_as_documented(oracle_built):
    """The claims of gs32_core.h about the reference: while nothing leaves int16 its 32-bit function in the -msse4.1 build gives
    the score of ns_global_gs16b (and mostly its CIGAR: the 4-lane stripe breaks some ties differently), the default build of
    the same function (no -msse4.1) gives neither."""
    rng = np.random.default_rng(99)
    from dpgen import make_task
    P = refbind.DpParams(refbind.mapping_matrix(23))
    score41 = cigar41 = score2 = 0
    for _ in range(40):
        nt, aa = make_task(rng)
        a, b = refbind.ref_nasw(nt, aa, P, 1), refbind.ref_gs32(nt, aa, P, 1)
        score41 += b[2] == a[2]
        cigar41 += b[3] == a[3]
        score2 += refbind.ref_gs32(nt, aa, P, 1, lib=refbind.ref())[2] == a[2]
    assert score41 == 40 and cigar41 >= 30 and score2 == 0


def test_tiny_windows(oracle_built):
    P = refbind.DpParams(refbind.mapping_matrix(23))
    for nl in (0, 1, 2):
        sc, cig = model_gs32(bytes([0] * nl), b"MK", P, 1)
        assert sc == GS32_NEG
