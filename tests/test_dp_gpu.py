"""Operator-level parity on the GPU: mpa_dp_run() (HIP kernels, through the C ABI) vs the oracle, bit-exact
on (nt_len, aa_len, score, CIGAR) for all three modes of ns_global_gs16b (nasw.h:135)."""
import numpy as np
import pytest
import miniprot_amd as mpa
import refbind
from dpgen import make_task
from dputil import build_workload, oracle_eval, dpopt_from_params, compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mpa.Context(0)
    yield c
    c.close()


def run_case(ctx, pairs, P, rng, modes=("cigar", "left", "right")):
    contigs, queries, tasks, meta = build_workload(pairs, rng, modes=modes, io=P.io)
    idx = mpa.Index.from_nt4(contigs)
    idx.to_device(ctx)
    rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
    expect = oracle_eval(pairs, meta, P)
    bad, msg = compare(rst, cig, expect, meta, pairs)
    idx.close()
    assert not bad, "%d/%d DP calls differ from the oracle\n%s" % (len(bad), len(tasks), msg)
    return len(tasks)


def test_mixed_small(ctx, oracle_built):
    rng = np.random.default_rng(11)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = [make_task(rng) for _ in range(600)]
    run_case(ctx, pairs, P, rng)


def test_shape_classes(ctx, oracle_built):
    """al right at the lane-group boundaries of the kernels (8,16,17,32,33,64,65,128,129,250)."""
    rng = np.random.default_rng(12)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = []
    for al in (1, 2, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 57, 63, 64, 65, 72, 120, 128, 129, 200, 250, 257, 300, 500, 513, 700, 1000):
        for _ in range(3):
            pairs.append(make_task(rng, al=al, p_indel=0.0))
    run_case(ctx, pairs, P, rng)


def test_long_windows(ctx, oracle_built):
    """extension windows with long flanks/introns (thousands of rows), as in mp_align (align.c:259-267)."""
    rng = np.random.default_rng(13)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = [make_task(rng, max_intron=6000, flank=5000) for _ in range(60)]
    run_case(ctx, pairs, P, rng)


def test_xdrop_and_params(ctx, oracle_built):
    """non-default scoring: small x-drop (so the extension really stops), other gap/intron/frameshift costs,
    the mammalian splice model, no length penalty."""
    rng = np.random.default_rng(14)
    for (go, ge, io, fs, xd, eb, sp, coef) in [(11, 1, 29, 23, 20, 5, (8, 15, 21, 30, 0, 0), 0.5),
                                                (5, 2, 40, 17, 50, 0, (8, 15, 21, 30, 4, 4), 1.0),
                                                (13, 1, 60, 10, 100, 10, (0, 0, 0, 0, 0, 0), 0.0)]:
        P = refbind.DpParams(refbind.mapping_matrix(fs), go=go, ge=ge, io=io, fs=fs, xdrop=xd, end_bonus=eb, sp=sp, ie_coef=coef)
        pairs = [make_task(rng, flank=300) for _ in range(150)]
        run_case(ctx, pairs, P, rng)


def test_gap_extension_and_frameshift_penalties_above_255(ctx, oracle_built):
    """-E / -F of the reference's command line accept any integer (main.c:133,136); above 255 the penalties do not fit the byte the
    row records give them and the run is swept by the kernels that read that byte as a stop flag (glob_cands<K, true>, k_ext_huge).
    Checked against the oracle and, where it is built, against the reference's own ns_global_gs16b."""
    rng = np.random.default_rng(21)
    for (go, ge, fs) in [(11, 300, 23), (11, 1, 400), (7, 260, 1000), (11, 2, 256), (9, 260, 100)]:
        P = refbind.DpParams(refbind.mapping_matrix(min(fs, 127)), go=go, ge=ge, io=29, fs=fs, xdrop=100, end_bonus=5)
        pairs = [make_task(rng, flank=200) for _ in range(60)] + [make_task(rng, al=al, p_indel=0.0) for al in (8, 33, 70, 130, 300)]
        pairs += [make_task(rng, max_intron=3000, flank=2000) for _ in range(20)]
        contigs, queries, tasks, meta = build_workload(pairs, rng, io=P.io)
        idx = mpa.Index.from_nt4(contigs)
        idx.to_device(ctx)
        rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
        idx.close()
        bad, msg = compare(rst, cig, oracle_eval(pairs, meta, P), meta, pairs)
        assert not bad, "go %d ge %d fs %d: %d/%d DP calls differ from the oracle\n%s" % (go, ge, fs, len(bad), len(tasks), msg)
        if refbind.have_ref():
            expect = []
            for k, fl, io in meta:
                nt, aa = pairs[k]
                PP = refbind.DpParams(P.mat, go=P.go, ge=P.ge, io=io, fs=P.fs, xdrop=P.xdrop, end_bonus=P.end_bonus, sp=P.sp, sp_null_bonus=P.sp_null_bonus, ie_coef=P.ie_coef)
                expect.append(refbind.ref_nasw(nt, aa, PP, fl))
            bad, msg = compare(rst, cig, expect, meta, pairs)
            assert not bad, "go %d ge %d fs %d: %d/%d DP calls differ from the REFERENCE\n%s" % (go, ge, fs, len(bad), len(tasks), msg)


def test_traceback_rounds(ctx, oracle_built, monkeypatch):
    """force several k_glob rounds by shrinking the traceback budget"""
    monkeypatch.setenv("MPA_TB_BUDGET_MB", "1")
    c2 = mpa.Context(0)
    rng = np.random.default_rng(15)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = [make_task(rng, max_intron=2000) for _ in range(200)]
    run_case(c2, pairs, P, rng, modes=("cigar",))
    c2.close()


def test_extension_wider_than_1024_columns(ctx, oracle_built):
    """maximum sizes: extension calls beyond the 1024 columns of k_ext_wide<16> go through k_ext_huge (the traceback kernel's
    int32 arithmetic, 64-column blocks swept one after the other) + k_ext_replay; global calls of that width use the
    block-major traceback class.  Mixed with ordinary calls in one batch."""
    rng = np.random.default_rng(16)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = []
    for al in (1025, 1030, 1100, 1600, 2500):
        pairs.append(make_task(rng, al=al, p_indel=0.002, flank=int(rng.choice([30, 400]))))
    pairs += [make_task(rng) for _ in range(40)]
    run_case(ctx, pairs, P, rng)
    # x-drop that really stops a wide extension, and a different end bonus
    P2 = refbind.DpParams(refbind.mapping_matrix(23), xdrop=30, end_bonus=11)
    run_case(ctx, [make_task(rng, al=1300, p_indel=0.01, flank=2000) for _ in range(3)], P2, rng)


def test_split_extension_calls_retry_and_load(oracle_built):
    """257..1024-column extension calls are swept by 2 / 4 workgroups with a hand-off through HBM (k_ext_wide_split).
    (a) A hand-off that times out makes mpa_dp_run() repeat the round with those calls on the one-wave path: forced here
    (MPA_TEST_HANDOFF_FAIL is read once per process, so the forced failure runs in a child), results must not change.
    (b) Four contexts sweeping such calls at the same time from four host threads (oversubscribed queues) agree with the oracle."""
    import os, subprocess, sys, threading
    code = ("import numpy as np, sys; sys.path.insert(0, 'tests'); import conftest, miniprot_amd as mpa, refbind\n"
            "from dpgen import make_task; from dputil import build_workload, oracle_eval, dpopt_from_params, compare\n"
            "rng = np.random.default_rng(21); P = refbind.DpParams(refbind.mapping_matrix(23))\n"
            "pairs = [make_task(rng, al=al, p_indel=0.002, flank=600) for al in (300, 513, 700, 1000, 1024, 40, 90)]\n"
            "contigs, queries, tasks, meta = build_workload(pairs, rng, modes=('left', 'right'), io=P.io)\n"
            "ctx = mpa.Context(0); idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)\n"
            "rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)\n"
            "bad, msg = compare(rst, cig, oracle_eval(pairs, meta, P), meta, pairs)\n"
            "assert not bad, msg\n"
            "print('retries', ctx.handoff_retries())\n")
    env = dict(os.environ, MPA_TEST_HANDOFF_FAIL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "retries 1" in out.stdout, out.stdout + out.stderr
    # (b)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    errs = []

    def worker(seed):
        try:
            c = mpa.Context(0)
            rng = np.random.default_rng(seed)
            pairs = [make_task(rng, al=int(rng.choice([513, 600, 800, 1000, 1024])), p_indel=0.002, flank=int(rng.choice([200, 3000]))) for _ in range(10)]
            pairs += [make_task(rng) for _ in range(60)]
            for _ in range(2):
                run_case(c, pairs, P, rng, modes=("left", "right"))
            assert c.handoff_retries() == 0
            c.close()
        except BaseException as e:                                   # noqa: BLE001 (reported in the main thread)
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(100 + k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_extension_that_saturates_int16(ctx, oracle_built):
    """scores beyond 32767: the reference's int16 arithmetic saturates; k_ext_huge clamps every operation the same way"""
    rng = np.random.default_rng(17)
    P = refbind.DpParams(refbind.mapping_matrix(23))
    pairs = [make_task(rng, al=3300, p_indel=0.001, flank=50),              # cannot use the packed kernels (al * max score > 32000), no saturation yet
             make_task(rng, al=6800, p_indel=0.0, p_sub=0.0, p_intron=0.0, p_fs=0.0, p_n=0.0, flank=40)]     # identical: the score runs into the int16 ceiling
    run_case(ctx, pairs, P, rng, modes=("right", "left"))


def test_splice_score_track(ctx, oracle_built, tmp_path):
    """--spsc at the operator level: the track resident next to the genome, applied by k_prep_rows, against the oracle fed
    the same ss[] bytes per call (all three modes, both strands, with and without MPA_F_SS_SKIP0)"""
    import golden
    from dputil import build_workload, dpopt_from_params, compare
    rng = np.random.default_rng(18)
    P = refbind.DpParams(refbind.mapping_matrix(23), io=39, sp_null_bonus=-7)
    pairs = [make_task(rng, max_intron=600, flank=int(rng.choice([20, 200]))) for _ in range(160)]
    contigs, queries, tasks, meta = build_workload(pairs, rng, io=39, io_alt=29)
    tasks["flag"] |= np.where((tasks["flag"] == mpa.F_EXT_LEFT) & (rng.random(len(tasks)) < 0.5), 8, 0).astype(tasks["flag"].dtype)
    idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
    mo = mpa.default_mapopt()
    idx.set_spsc(golden.write_spsc({"spsc": 77}, contigs, str(tmp_path / "s.tsv")), mo, keep_io=True)
    idx.to_device(ctx)
    dp = dpopt_from_params(P)
    dp.sp_null_bonus = P.sp_null_bonus
    rst, cig = mpa.dp_run(ctx, idx, dp, queries, tasks)
    expect = []
    for t, (k, fl, io) in zip(tasks, meta):
        st, nl, vid = int(t["nt_off"]), int(t["nl"]), int(t["vid"])
        ss = (idx.get_spsc(vid, st - 1, st + nl)[1:] if st > 0 else idx.get_spsc(vid, 0, nl)).copy()
        if int(t["flag"]) & 8:
            ss[0] = 0xff
        PP = refbind.DpParams(P.mat, io=io, sp_null_bonus=P.sp_null_bonus)
        nt = bytes(idx.get_nt(vid, st, st + nl))
        expect.append(refbind.ora_nasw(nt, pairs[k][1], PP, fl, ss))
    bad, msg = compare(rst, cig, expect, meta, pairs)
    idx.close()
    assert not bad, "%d/%d DP calls differ from the oracle\n%s" % (len(bad), len(tasks), msg)


def test_antidiagonal_prototype_equals_the_oracle(oracle_built):
    """dp_antidiag.hip: the anti-diagonal mapping of the score-only extension sweep (lane j on row t - j; the mapping BASELINE.json's
    north_star names), switched in for the 32-column class with mpa_dbg_antidiag(): (nt_len, aa_len, score) of left and right
    extension calls of 17..32 columns -- alone in a wave and packed four to a wave, windows shorter than the diagonal's ramp,
    long windows, a small x-drop, non-default scores -- against the oracle."""
    import ctypes as C
    c2 = mpa.Context(0)
    mpa.lib().mpa_dbg_antidiag.argtypes = [C.c_void_p, C.c_int]
    mpa.lib().mpa_dbg_antidiag(c2.h, 1)
    rng = np.random.default_rng(41)
    n = 0
    for (go, ge, io, fs, xd, eb, sp, coef) in [(11, 1, 29, 23, 100, 5, (8, 15, 21, 30, 0, 0), 0.5),
                                                (11, 1, 29, 23, 15, 5, (8, 15, 21, 30, 0, 0), 0.5),
                                                (5, 2, 40, 17, 50, 0, (8, 15, 21, 30, 4, 4), 1.0)]:
        P = refbind.DpParams(refbind.mapping_matrix(fs), go=go, ge=ge, io=io, fs=fs, xdrop=xd, end_bonus=eb, sp=sp, ie_coef=coef)
        pairs = []
        for al in (17, 18, 23, 24, 25, 31, 32):
            for k in range(10):
                pairs.append(make_task(rng, al=al, flank=int(rng.choice([1, 5, 40, 300, 3000])), p_indel=0.0 if k < 7 else 0.02))
        pairs += [make_task(rng, al=int(rng.integers(17, 33)), max_intron=6000, flank=5000) for _ in range(12)]
        pairs += [make_task(rng, al=20, flank=1, p_indel=0.0)]
        n += run_case(c2, pairs, P, rng, modes=("left", "right"))
        n += run_case(c2, pairs[:3], P, rng, modes=("right",))     # a wave with empty slots
    assert c2.dp_stats()["launches_ext"] >= 1
    assert n > 400
    c2.close()


def test_checkpointed_traceback(oracle_built, monkeypatch):
    """Traceback calls of up to 128 columns and many rows take the checkpointed path (dp_device.h): the packed sweep keeps four
    extension bits per cell and a checkpoint per 96 rows, k_walk follows deletion / intron runs on the bits and recomputes the
    traceback words of a block only where the path needs them.  CIGARs and scores against the oracle for every lane class (16 /
    32 / 64 lanes per call, full and partly filled waves; 65..128 columns: one call per wave, column c + 64 in the high half of lane c), paths that change state right at block boundaries (several block
    phases through the flank), introns of all phases, frameshifts, indels, N runs; and with the row threshold lowered so that
    calls of a single block and of two blocks go the same way."""
    P = refbind.DpParams(refbind.mapping_matrix(23))
    for lite_min, seed in (("384", 31), ("3", 32), ("100", 33)):
        monkeypatch.setenv("MPA_DP_LITE_MIN", lite_min)
        rng = np.random.default_rng(seed)
        pairs = []
        for al in (1, 5, 8, 9, 16, 17, 23, 24, 25, 31, 32, 33, 40, 47, 48, 49, 56, 57, 63, 64, 65, 66, 72, 73, 90, 104, 120, 121, 127, 128):
            for flank in (3, 90, 200, 700):
                pairs.append(make_task(rng, al=al, max_intron=int(rng.choice([300, 1500, 5000])), flank=flank, p_intron=0.15))
        pairs += [make_task(rng, max_intron=2500, flank=400, p_fs=0.05, p_indel=0.06, p_n=0.01) for _ in range(150)]
        pairs += [make_task(rng, al=int(rng.integers(1, 129)), max_intron=20000, flank=50, p_intron=0.1) for _ in range(16)]
        pairs += [make_task(rng, al=int(rng.integers(65, 129)), max_intron=2500, flank=400, p_fs=0.05, p_indel=0.06, p_n=0.01) for _ in range(60)]
        c2 = mpa.Context(0)
        run_case(c2, pairs, P, rng, modes=("cigar",))
        c2.close()
    # other scoring: the bits follow the penalties (gap extension 2, intron open 40, mammalian splice model)
    monkeypatch.setenv("MPA_DP_LITE_MIN", "200")
    rng = np.random.default_rng(34)
    P2 = refbind.DpParams(refbind.mapping_matrix(17), go=5, ge=2, io=40, fs=17, xdrop=50, end_bonus=0, sp=(8, 15, 21, 30, 4, 4), ie_coef=1.0)
    c2 = mpa.Context(0)
    run_case(c2, [make_task(rng, max_intron=3000, flank=300) for _ in range(200)], P2, rng)
    c2.close()
