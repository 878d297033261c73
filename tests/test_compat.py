"""The reference's own interface (include/miniprot.h, include/nasw.h) as exported by libmpamd.so: record layouts,
tables, and the reference's own programs -- example.c and main.c, compiled UNCHANGED from /root/reference by
examples/Makefile -- relinked against this library.  CPU part: layouts, genetic codes, index files.  GPU part: the
relinked programs and ns_global_gs16b() produce the reference's bytes."""
import ctypes as C
import hashlib
import os
import subprocess
import numpy as np
import pytest
import miniprot_amd as mpa
import refbind
import golden
import gen_synth

ROOT = refbind.ROOT
RELINK = os.path.join(ROOT, "examples", "_relink")
CLI, LITE = os.path.join(RELINK, "miniprot"), os.path.join(RELINK, "miniprot-lite")
REF_LITE = os.path.join(ROOT, "oracle", "_ref", "miniprot-lite")

LAYOUT_PROG = r"""
#include <stdio.h>
#include <stddef.h>
#include "miniprot.h"
#include "nasw.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
	S(mp_idxopt_t); S(mp_mapopt_t); S(mp_ctg_t); S(mp_ntdb_t); S(mp_idx_t); S(mp_extra_t); S(mp_feat_t); S(mp_reg1_t); S(ns_opt_t); S(ns_rst_t);
	O(mp_mapopt_t, mini_batch_size); O(mp_mapopt_t, max_ext); O(mp_mapopt_t, chn_coef_log); O(mp_mapopt_t, io_end); O(mp_mapopt_t, sp_scale);
	O(mp_mapopt_t, gff_prefix); O(mp_mapopt_t, mat);
	O(mp_ntdb_t, l_seq); O(mp_ntdb_t, seq); O(mp_ntdb_t, ctg); O(mp_ntdb_t, spsc); O(mp_idx_t, nt); O(mp_idx_t, ki); O(mp_idx_t, kb);
	O(mp_extra_t, dist_start); O(mp_extra_t, cigar); O(mp_feat_t, phase); O(mp_feat_t, donor);
	O(mp_reg1_t, n_feat); O(mp_reg1_t, hash); O(mp_reg1_t, vs); O(mp_reg1_t, a); O(mp_reg1_t, feat); O(mp_reg1_t, p);
	O(ns_opt_t, sp); O(ns_opt_t, ie_coef); O(ns_opt_t, sc); O(ns_opt_t, codon); O(ns_rst_t, score); O(ns_rst_t, cigar);
	printf("flags %d %d %d %d %s %s\n", MP_F_GFF, MP_F_NO_CS, NS_F_EXT_RIGHT, NS_CIGAR_V, NS_CIGAR_STR, MP_VERSION);
	return 0;
}
"""


def _layout(tmp_path, incdirs, tag):
    src = tmp_path / ("layout_%s.c" % tag)
    src.write_text(LAYOUT_PROG)
    exe = str(tmp_path / ("layout_" + tag))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror"] + ["-I" + d for d in incdirs] + [str(src), "-o", exe], check=True)
    return subprocess.run([exe], capture_output=True, text=True, check=True).stdout


def test_reference_records_have_the_reference_layout(tmp_path):
    """include/miniprot.h + include/nasw.h compile as plain C99 and describe the reference's ABI: the sizes SURVEY.md 8(b)
    measured on the reference (104/48/56/656/96/32/20) and, where /root/reference is present, every probed offset"""
    ours = _layout(tmp_path, [os.path.join(ROOT, "include")], "ours")
    d = dict(l.rsplit(" ", 1) for l in ours.splitlines() if not l.startswith("flags"))
    assert (d["mp_reg1_t"], d["mp_extra_t"], d["mp_feat_t"], d["mp_mapopt_t"], d["ns_opt_t"], d["ns_rst_t"], d["mp_idxopt_t"]) == \
        ("104", "48", "56", "656", "96", "32", "20")
    if os.path.isdir("/root/reference"):
        assert ours == _layout(tmp_path, ["/root/reference"], "ref")


@pytest.mark.skipif(not refbind.have_ref(), reason="compiled reference not available")
def test_every_genetic_code_of_the_reference():
    """ns_make_tables(): same return code and the same codon tables as nasw-tab.c for every NCBI code 0..34"""
    L, R = mpa.lib(), refbind.ref()
    def tab(lib, name, n):
        return bytes((C.c_uint8 * n).in_dll(lib, name))
    try:
        for code in range(-1, 36):
            a, b = L.ns_make_tables(code), R.ns_make_tables(code)
            assert a == b, "return code for table %d: %d vs %d" % (code, a, b)
            if a == 0:
                for name, n in (("ns_tab_codon", 64), ("ns_tab_codon13", 64), ("ns_tab_nt4", 256), ("ns_tab_aa20", 256), ("ns_tab_aa13", 256), ("ns_tab_a2r", 22)):
                    assert tab(L, name, n) == tab(R, name, n), "%s differs for genetic code %d" % (name, code)
        assert bytes((C.c_int8 * 484).in_dll(L, "ns_mat_blosum62")) == bytes((C.c_int8 * 484).in_dll(R, "ns_mat_blosum62"))
    finally:
        L.ns_make_tables(1), R.ns_make_tables(1)


def _multi_contig_genome(path, seed=7):
    """contigs of very different sizes (one shorter than a block, one shorter than a k-mer), N runs, lower case, IUPAC codes"""
    rng = np.random.default_rng(seed)
    contigs, prots, names = gen_synth.generate(1_200_000, 3, 30, seed, n_frac=0.05, paralog_frac=0.02)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            s = lut[g].copy()
            if i == 1:
                s[1000:3000] |= 0x20                            # soft-masked stretch
                s[5000:5010] = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)
            f.write(b">ctg%d some comment\n" % i)
            for p in range(0, len(s), 60 + 13 * i):              # different line widths per contig
                f.write(s[p:p + 60 + 13 * i].tobytes() + b"\n")
        f.write(b">tiny\nACGTACGTAC\n>short_block\n" + lut[rng.integers(0, 4, 200)].tobytes() + b"\n>empty\n\n")
    return prots, names


@pytest.mark.skipif(not (os.path.exists(CLI) and os.path.exists(refbind.REF_BIN)), reason="relinked / reference programs not built")
def test_relinked_cli_writes_the_reference_index(tmp_path):
    """`miniprot -d` of the reference's main.c linked with libmpamd.so: the .mpi of a multi-contig genome with N runs,
    tiny and empty contigs is the reference's file byte for byte (index.c:189-202), also with non-default -k/-M/-b/-L"""
    fa = str(tmp_path / "g.fa")
    _multi_contig_genome(fa)
    for k, flags in enumerate(([], ["-k5", "-M0", "-b6", "-L20"])):
        a, b = str(tmp_path / ("a%d.mpi" % k)), str(tmp_path / ("b%d.mpi" % k))
        subprocess.run([CLI, "-t4"] + flags + ["-d", a, fa], check=True, capture_output=True)
        subprocess.run([refbind.REF_BIN, "-t4"] + flags + ["-d", b, fa], check=True, capture_output=True)
        assert open(a, "rb").read() == open(b, "rb").read()
    # and the library reads the reference's file back into the same bytes
    idx = mpa.Index.restore(str(tmp_path / "b0.mpi"))
    idx.dump(str(tmp_path / "c.mpi"))
    assert open(str(tmp_path / "c.mpi"), "rb").read() == open(str(tmp_path / "b0.mpi"), "rb").read()


@pytest.mark.skipif(not os.path.exists(CLI), reason="relinked programs not built")
def test_relinked_cli_fails_loudly_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([CLI, golden.path("DPP3-hs.gen.fa.gz"), golden.path("DPP3-mm.pep.fa.gz")], capture_output=True)
    assert r.returncode != 0 and b"no usable HIP device" in r.stderr and r.stdout == b""


# ------------------------------------------------------------------------------------------------ GPU
def _run(cmd):
    r = subprocess.run(cmd, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_relinked_example_prints_the_reference_lines():
    """/root/reference/example.c, unchanged, compiled against include/miniprot.h and linked with -lmpamd: mp_idx_load +
    mp_map per query + the caller freeing ->p -- same lines as the reference's miniprot-lite on DPP3"""
    args = [golden.path("DPP3-hs.gen.fa.gz"), golden.path("DPP3-mm.pep.fa.gz")]
    assert _run([LITE] + args) == _run([REF_LITE] + args)


@pytest.mark.gpu
def test_relinked_cli_is_a_drop_in(tmp_path):
    """the reference's main.c linked with libmpamd.so IS miniprot's command line: config 1 with the md5 of SURVEY.md 8(c),
    option parsing -> mp_map_file() -> stdout for a handful of flag sets, compared with the reference binary run live"""
    g, p = golden.path("DPP3-hs.gen.fa.gz"), golden.path("DPP3-mm.pep.fa.gz")
    out = _run([CLI, "-t1", g, p])
    assert hashlib.md5(out).hexdigest() == "74fd00200bda6c03380bb3062fb5178b"
    assert hashlib.md5(_run([CLI, "--gff", g, p])).hexdigest() == "d08d9caa27b8bd896ee4f14824597c54"
    fa = str(tmp_path / "g.fa")
    prots, names = _multi_contig_genome(fa)
    faa = str(tmp_path / "p.faa")
    gen_synth.write_fasta_aa(faa, prots, names)
    mpi = str(tmp_path / "g.mpi")
    _run([CLI, "-t4", "-d", mpi, fa])
    # (-E above 255: the reference takes any integer, main.c:133 -- the run is swept by the kernels that keep a stop flag in the row
    # records' byte; -F is held to an int8 by the reference itself, options.c:26, so a frameshift penalty above 255 can only arrive
    # through the operator, ns_opt_t: tests/test_dp_gpu.py)
    for flags in (["-u"], ["-u", "-I", "--gtf"], ["-u", "--gff", "--outs=0.5", "-N", "50", "-K", "3k"], ["-u", "-S"], ["-u", "-T", "4", "--aln", "--trans"],
                  ["-u", "-E", "300"], ["-u", "-E", "260", "-O", "9"]):     # (with -F 60..100 next to -E 260 the reference aborts in its own mp_extra_cal assertion -- and so does this)
        src = [fa] if "-T" in flags else [mpi]                      # (-T changes the index: build it from the FASTA)
        assert _run([CLI, "-t8"] + flags + src + [faa]) == _run([refbind.REF_BIN, "-t8"] + flags + src + [faa]), " ".join(flags)


@pytest.mark.gpu
def test_relinked_example_on_a_synthetic_genome(tmp_path):
    fa = str(tmp_path / "g.fa")
    prots, names = _multi_contig_genome(fa, seed=11)
    faa = str(tmp_path / "p.faa")
    gen_synth.write_fasta_aa(faa, prots, names)
    assert _run([LITE, fa, faa]) == _run([REF_LITE, fa, faa])


@pytest.mark.gpu
def test_ns_global_gs16b_on_a_host_buffer():
    """the operator under its reference name (nasw.h:131): ASCII or 0..4 nucleotides in host memory, ns_opt_t in, ns_rst_t
    out with a malloc'd CIGAR -- against the compiled reference on the same random calls (global, both extensions, ss[])"""
    from dpgen import make_task, make_ss
    L, R = mpa.lib(), refbind.ref()
    L.ns_global_gs16b.argtypes = R.ns_global_gs16b.argtypes
    L.ns_global_gs16b.restype = None
    rng = np.random.default_rng(23)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    n_checked = 0
    for it in range(60):
        nt, aa = make_task(rng)
        flag = (1, 2, 4)[it % 3]                                     # NS_F_CIGAR (global) / NS_F_EXT_LEFT / NS_F_EXT_RIGHT
        if it % 5 == 0:
            nt = bytes(b"ACGTN"[c] for c in nt)                      # ASCII instead of 0..4 codes
        ss = np.frombuffer(make_ss(rng, len(nt), 0.05), dtype=np.uint8).copy() if it % 4 == 1 else None
        res = []
        for lib in (L, R):
            o = refbind.NsOpt()
            lib.ns_opt_init(C.byref(o))
            o.flag = flag
            o.io = 29 if it % 2 else 19
            r = refbind.NsRst()
            lib.ns_global_gs16b(None, bytes(nt), len(nt), bytes(aa), len(aa), C.byref(o), ss.ctypes.data if ss is not None else None, C.byref(r))
            cig = [r.cigar[k] for k in range(r.n_cigar)]
            if r.cigar:
                libc.free(r.cigar)
            res.append((r.nt_len, r.aa_len, r.score, cig))
        assert res[0] == res[1], "call %d (flag %d, nl %d, al %d)" % (it, flag, len(nt), len(aa))
        n_checked += 1
    assert n_checked == 60


@pytest.mark.gpu
def test_ns_global_gs32b_on_the_gpu(oracle_built):
    """the 32-bit operator (nasw.h:129,132) on the device: every golden vector of the reference's -msse4.1 build (scores far
    beyond int16, other penalties, splice tracks, reversed preparation, tiny windows) through ns_global_gs32b -- one wavefront
    per call runs the source the CPU tests pin (gs32_core.h) -- and a batch of calls through mpa_dp_run32 against the same
    reference build (or, without it on the box, against the host model the CPU tests pinned to it)."""
    import golden
    from gs32util import model_gs32, gs32_cases
    L = mpa.lib()
    L.ns_global_gs32b.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(refbind.NsOpt), C.c_void_p, C.POINTER(refbind.NsRst)]
    L.ns_global_gs32b.restype = None
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    z = np.load(golden.path("gs32_vectors.npz"))
    n = len(z["flag"])
    nt_off, aa_off = np.concatenate([[0], np.cumsum(z["nt_len"])]), np.concatenate([[0], np.cumsum(z["aa_len"])])
    cg_off, ss_off = np.concatenate([[0], np.cumsum(z["cig_len"])]), np.concatenate([[0], np.cumsum(z["ss_len"])])
    for k in range(n):
        nt, aa = bytes(z["nt"][nt_off[k]:nt_off[k + 1]]), bytes(z["aa"][aa_off[k]:aa_off[k + 1]])
        ss = np.ascontiguousarray(z["ss"][ss_off[k]:ss_off[k + 1]]) if z["ss_len"][k] else None
        o = refbind.NsOpt()
        L.ns_opt_init(C.byref(o))
        mat = np.ascontiguousarray(z["mat"][k], dtype=np.int8)
        o.flag, o.sc = int(z["flag"][k]), mat.ctypes.data
        o.go, o.ge, o.io, o.fs = (int(x) for x in z["par"][k][:4])
        for i in range(6):
            o.sp[i] = int(z["par"][k][4 + i])
        r = refbind.NsRst()
        L.ns_global_gs32b(None, nt, len(nt), aa, len(aa), C.byref(o), ss.ctypes.data if ss is not None else None, C.byref(r))
        cig = [r.cigar[i] for i in range(r.n_cigar)]
        if r.cigar:
            libc.free(r.cigar)
        want = (len(nt), len(aa), int(z["score"][k]), [int(x) for x in z["cig"][cg_off[k]:cg_off[k + 1]]])
        assert (r.nt_len, r.aa_len, r.score, cig) == want, "vector %d (flag %d, nl %d, al %d)" % (k, o.flag, len(nt), len(aa))
    # a batch through mpa_dp_run32: one set of parameters, intron-open penalty per call
    from dputil import build_workload, dpopt_from_params
    rng = np.random.default_rng(77)
    mat = np.clip(refbind.mapping_matrix(23).astype(np.int32) * 11, -128, 127).astype(np.int8)
    P = refbind.DpParams(mat)
    from dpgen import make_task
    pairs = [make_task(rng) for _ in range(40)] + [make_task(rng, al=int(rng.integers(500, 1200)), p_sub=0.02, p_indel=0.004, p_intron=0.01) for _ in range(6)]
    contigs, queries, tasks, meta = build_workload(pairs, rng, modes=("cigar",), io=P.io)
    ctx = mpa.Context(0)
    idx = mpa.Index.from_nt4(contigs)
    idx.to_device(ctx)
    rst, pool = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks, wide32=True)
    n_big = 0
    for t, (nt, aa) in enumerate(pairs):
        if refbind.have_ref41():
            want = refbind.ref_gs32(nt, aa, P, 1)
        else:
            sc, cg = model_gs32(nt, aa, P, 1)
            want = (len(nt), len(aa), sc, cg)
        got = (int(rst[t]["nt_len"]), int(rst[t]["aa_len"]), int(rst[t]["score"]), [int(x) for x in pool[rst[t]["cigar_off"]:rst[t]["cigar_off"] + rst[t]["n_cigar"]]])
        assert got == want, "call %d (nl %d, al %d)" % (t, len(nt), len(aa))
        n_big += want[2] > 32767
    assert n_big >= 4
    idx.close(), ctx.close()


@pytest.mark.gpu
def test_index_built_on_the_gpu_is_the_reference_file(tmp_path):
    """mpa_idx_build_kmers_device (scan kernel + one radix sort of all k-mer keys + unique) against the host build, which the
    CPU tests pin to `miniprot -d`: the .mpi of the awkward multi-contig genome (tiny, empty, N-rich, soft-masked contigs), with
    default and non-default -k/-M/-b/-L, and of a 30 Mbp synthetic genome; mapping with the GPU-built index gives the same bytes"""
    fa = str(tmp_path / "g.fa")
    prots, names = _multi_contig_genome(fa)
    ctx = mpa.Context(0)
    for k, io in enumerate((None, (6, 20, 5, 0, 1), (8, 30, 6, 2, 1))):
        a = mpa.Index.read_fasta(fa, io)
        b = mpa.Index.read_fasta(fa, io)
        assert a.build_kmers(4, ctx) == "gpu" and b.build_kmers(4) == "host"
        a.dump(str(tmp_path / "a.mpi")), b.dump(str(tmp_path / "b.mpi"))
        assert open(str(tmp_path / "a.mpi"), "rb").read() == open(str(tmp_path / "b.mpi"), "rb").read(), io
        if io is None and os.path.exists(refbind.REF_BIN):
            subprocess.run([refbind.REF_BIN, "-t4", "-d", str(tmp_path / "r.mpi"), fa], check=True, capture_output=True)
            assert open(str(tmp_path / "a.mpi"), "rb").read() == open(str(tmp_path / "r.mpi"), "rb").read()
        a.close(), b.close()
    contigs, prots, names = gen_synth.generate(30_000_000, 3, 60, 21, n_frac=0.05, paralog_frac=0.02)
    a = mpa.Index.from_nt4(contigs, ["c1", "c2", "c3"])
    b = mpa.Index.from_nt4(contigs, ["c1", "c2", "c3"])
    assert a.build_kmers(8, ctx) == "gpu" and b.build_kmers(8) == "host"
    a.dump(str(tmp_path / "a.mpi")), b.dump(str(tmp_path / "b.mpi"))
    assert open(str(tmp_path / "a.mpi"), "rb").read() == open(str(tmp_path / "b.mpi"), "rb").read()
    mo = mpa.default_mapopt()
    mo.flag |= 4
    a.to_device(ctx)
    ours = b"".join(mpa.map_batches(ctx, a, mo, [mpa.Queries(prots, names)], 8))
    if os.path.exists(refbind.REF_BIN):
        faa = str(tmp_path / "p.faa")
        gen_synth.write_fasta_aa(faa, prots, names)
        assert ours == subprocess.run([refbind.REF_BIN, "-t8", "-u", str(tmp_path / "b.mpi"), faa], capture_output=True).stdout
    a.close(), b.close(), ctx.close()
