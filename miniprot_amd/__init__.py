"""miniprot_amd -- Python view of libmpamd.so, the MI355X-native drop-in for miniprot's hot path.

This module is plumbing only: it loads the C-ABI library (include/mpamd.h) with ctypes and offers thin
helpers for the tests and bench.py.  All compute lives in miniprot_amd/csrc (hand-written HIP for
gfx950 + host C++).  There is no Python or CPU fallback: if the shared library is missing this import
fails, and if no GPU is usable Context() raises.
"""
import ctypes as C
import os
import subprocess
import numpy as np

# one HIP stream per kernel class: give the runtime enough hardware queues (must be set before HIP initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPA_LIB_PATH") or os.path.join(_HERE, "libmpamd.so")   # (override: experiment builds)

F_CIGAR, F_EXT_LEFT, F_EXT_RIGHT = 1, 2, 4

DP_TASK = np.dtype([("nt_off", "<i8"), ("vid", "<i4"), ("nl", "<i4"), ("qid", "<i4"), ("aa_off", "<i4"),
                    ("al", "<i4"), ("flag", "<i4"), ("io", "<i4"), ("tag", "<i4")], align=True)
DP_RST = np.dtype([("nt_len", "<i4"), ("aa_len", "<i4"), ("score", "<i4"), ("n_cigar", "<i4"),
                   ("cigar_off", "<i8")], align=True)
HIT = np.dtype([("qid", "<i4"), ("id", "<i4"), ("parent", "<i4"), ("n_sub", "<i4"), ("subsc", "<i4"), ("cnt", "<i4"),
                ("n_exon", "<i4"), ("chn_sc", "<i4"), ("chn_sc_ungap", "<i4"), ("vid", "<u4"), ("qs", "<i4"),
                ("qe", "<i4"), ("vs", "<i8"), ("ve", "<i8"), ("has_aln", "<i4"), ("dp_score", "<i4"),
                ("dp_max", "<i4"), ("dp_max2", "<i4"), ("blen", "<i4"), ("n_fs", "<i4"), ("n_stop", "<i4"),
                ("dist_stop", "<i4"), ("dist_start", "<i4"), ("n_iden", "<i4"), ("n_plus", "<i4"),
                ("n_cigar", "<i4"), ("n_feat", "<i4"), ("cigar_off", "<i8"), ("feat_off", "<i8")], align=True)


class MapOpt(C.Structure):
    _fields_ = [("flag", C.c_uint32), ("mini_batch_size", C.c_int64), ("max_occ", C.c_int32), ("max_gap", C.c_int32),
                ("max_intron", C.c_int32), ("min_max_intron", C.c_int32), ("max_max_intron", C.c_int32),
                ("bw", C.c_int32), ("max_ext", C.c_int32), ("max_ava", C.c_int32), ("min_chn_cnt", C.c_int32),
                ("max_chn_max_skip", C.c_int32), ("max_chn_iter", C.c_int32), ("min_chn_sc", C.c_int32),
                ("chn_coef_log", C.c_float), ("mask_level", C.c_float), ("mask_len", C.c_int32),
                ("pri_ratio", C.c_float), ("out_sim", C.c_float), ("out_cov", C.c_float), ("best_n", C.c_int32),
                ("out_n", C.c_int32), ("kmer2", C.c_int32), ("go", C.c_int32), ("ge", C.c_int32), ("io", C.c_int32),
                ("fs", C.c_int32), ("io_end", C.c_int32), ("ie_coef", C.c_float), ("sp_model", C.c_int32),
                ("sp_null_bonus", C.c_int32), ("sp_max_bonus", C.c_int32), ("sp_scale", C.c_float),
                ("xdrop", C.c_int32), ("end_bonus", C.c_int32), ("asize", C.c_int32), ("gff_delim", C.c_int32),
                ("max_intron_flank", C.c_int32), ("gff_prefix", C.c_char_p), ("mat", C.c_int8 * 484)]


class DpOpt(C.Structure):
    _fields_ = [("go", C.c_int32), ("ge", C.c_int32), ("fs", C.c_int32), ("xdrop", C.c_int32),
                ("end_bonus", C.c_int32), ("sp", C.c_int32 * 6), ("ie_coef", C.c_float), ("mat", C.c_int8 * 484),
                ("sp_null_bonus", C.c_int32)]


class QBatch(C.Structure):
    _fields_ = [("n_seq", C.c_int32), ("seqs", C.c_char_p), ("q_off", C.POINTER(C.c_int64))]


class DpStats(C.Structure):
    _fields_ = [("n_ext", C.c_int64), ("n_glob", C.c_int64), ("cells_ext", C.c_int64), ("cells_glob", C.c_int64),
                ("rows_prep", C.c_int64), ("alg_bytes_ext", C.c_int64), ("alg_bytes_glob", C.c_int64),
                ("ms_prep", C.c_double), ("ms_ext", C.c_double), ("ms_glob", C.c_double),
                ("ms_backtrack", C.c_double), ("ms_total", C.c_double), ("launches_ext", C.c_int32),
                ("launches_glob", C.c_int32), ("cells_ext_round", C.c_int64), ("cells_glob_round", C.c_int64),
                ("ms_round", C.c_double), ("launches_round", C.c_int32), ("pad_", C.c_int32), ("ms_round_union", C.c_double),
                ("n_ckpt", C.c_int64), ("cells_ckpt", C.c_int64), ("walk_blocks", C.c_int64)]


def build(verbose=False):
    """Compile libmpamd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libmpamd.so failed:\n" + (r.stdout or "") + (r.stderr or ""))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no Python/CPU fallback for the HIP kernels)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.mpa_last_error.restype = C.c_char_p
        L.mpa_version.restype = C.c_char_p
        L.mpa_ctx_create.restype = C.c_void_p
        L.mpa_ctx_create.argtypes = [C.c_int]
        L.mpa_ctx_destroy.argtypes = [C.c_void_p]
        L.mpa_idx_restore.restype = C.c_void_p
        L.mpa_idx_restore.argtypes = [C.c_char_p]
        L.mpa_idx_from_nt4.restype = C.c_void_p
        L.mpa_idx_from_nt4.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p]
        L.mpa_idx_dump.argtypes = [C.c_char_p, C.c_void_p]
        L.mpa_idx_build_kmers.argtypes = [C.c_void_p, C.c_int]
        L.mpa_idx_build_kmers_device.argtypes = [C.c_void_p, C.c_void_p]
        L.mpa_idx_read_fasta.restype = C.c_void_p
        L.mpa_idx_read_fasta.argtypes = [C.c_char_p, C.c_void_p]
        L.mpa_idx_destroy.argtypes = [C.c_void_p]
        L.mpa_idx_n_ctg.argtypes = [C.c_void_p]
        L.mpa_idx_ctg_len.restype = C.c_int64
        L.mpa_idx_ctg_len.argtypes = [C.c_void_p, C.c_int32]
        L.mpa_idx_ctg_name.restype = C.c_char_p
        L.mpa_idx_ctg_name.argtypes = [C.c_void_p, C.c_int32]
        L.mpa_idx_genome_len.restype = C.c_int64
        L.mpa_idx_genome_len.argtypes = [C.c_void_p]
        L.mpa_idx_get_nt.restype = C.c_int64
        L.mpa_idx_get_nt.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        L.mpa_idx_to_device.argtypes = [C.c_void_p, C.c_void_p]
        L.mpa_mapopt_init.argtypes = [C.POINTER(MapOpt)]
        L.mpa_mapopt_set_fs.argtypes = [C.POINTER(MapOpt), C.c_int32]
        L.mpa_mapopt_set_max_intron.argtypes = [C.POINTER(MapOpt), C.c_int64]
        L.mpa_dpopt_from_mapopt.argtypes = [C.POINTER(MapOpt), C.POINTER(DpOpt)]
        L.mpa_dp_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(DpOpt), C.POINTER(QBatch), C.c_int64, C.c_void_p,
                                 C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int64)]
        L.mpa_dp_last_stats.argtypes = [C.c_void_p, C.POINTER(DpStats)]
        L.mpa_dp_total_stats.argtypes = [C.c_void_p, C.POINTER(DpStats), C.c_int]
        if hasattr(L, "mpa_dp_handoff_retries"):                      # (absent from older builds loaded through MPA_LIB_PATH)
            L.mpa_dp_handoff_retries.argtypes = [C.c_void_p]
            L.mpa_dp_handoff_retries.restype = C.c_int64
        # (an older build loaded through MPA_LIB_PATH may lack the diagnostics below: the accessors then return 0 / empty)
        if hasattr(L, "mpa_device_bytes"):
            L.mpa_device_bytes.restype = C.c_int64
        if hasattr(L, "mpa_pool_growths"):
            L.mpa_pool_growths.restype = C.c_int64
        if hasattr(L, "mpa_stage_clocks"):
            L.mpa_stage_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mpa_free.argtypes = [C.c_void_p]
        for name, res, args in [
            ("mpa_batch_begin", C.c_void_p, [C.c_void_p, C.POINTER(MapOpt), C.POINTER(QBatch), C.c_int]),
            ("mpa_batch_dp_tasks", C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(DpOpt)]),
            ("mpa_batch_dp_results", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
            ("mpa_batch_finish", C.c_void_p, [C.c_void_p]),
            ("mpa_map_batch", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MapOpt), C.POINTER(QBatch), C.c_int,
                                        C.POINTER(C.c_void_p)]),
            ("mpa_result_n_hit", C.c_int64, [C.c_void_p]),
            ("mpa_result_hits", C.c_void_p, [C.c_void_p]),
            ("mpa_result_hit_off", C.POINTER(C.c_int64), [C.c_void_p]),
            ("mpa_result_cigars", C.POINTER(C.c_uint32), [C.c_void_p]),
            ("mpa_result_destroy", None, [C.c_void_p]),
            ("mpa_idx_set_spsc", C.c_int64, [C.c_void_p, C.c_char_p, C.POINTER(MapOpt), C.c_int]),
            ("mpa_idx_get_spsc", C.c_int64, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
            ("mpa_map_batches", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MapOpt), C.c_int32, C.POINTER(QBatch),
                                          C.POINTER(C.POINTER(C.c_char_p)), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
            ("mpa_format_output", C.c_int64, [C.c_void_p, C.POINTER(MapOpt), C.POINTER(QBatch), C.POINTER(C.c_char_p),
                                              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
            ("mpa_result_n_output", C.c_int64, [C.POINTER(MapOpt), C.POINTER(QBatch), C.c_void_p]),
            ("mpa_format_paf", C.c_int64, [C.c_void_p, C.POINTER(MapOpt), C.POINTER(QBatch), C.POINTER(C.c_char_p),
                                           C.c_void_p, C.POINTER(C.c_void_p)]),
        ]:
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def last_error():
    return lib().mpa_last_error().decode()


class MpaError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise MpaError("libmpamd error %d: %s" % (rc, last_error()))


class Context:
    """A HIP device context (one per GPU / per process rank)."""

    def __init__(self, device=0):
        self.h = lib().mpa_ctx_create(device)
        if not self.h:
            raise MpaError("cannot create a device context: " + last_error())

    def close(self):
        if self.h:
            lib().mpa_ctx_destroy(self.h)
            self.h = None

    def dp_stats(self, total=False, reset=False):
        st = DpStats()
        if total:
            lib().mpa_dp_total_stats(self.h, C.byref(st), 1 if reset else 0)
        else:
            lib().mpa_dp_last_stats(self.h, C.byref(st))
        return {k: getattr(st, k) for k, _ in DpStats._fields_}

    def handoff_retries(self):
        """DP rounds repeated because a workgroup hand-off of a split extension call timed out (mpa_dp_handoff_retries)."""
        return int(lib().mpa_dp_handoff_retries(self.h)) if hasattr(lib(), "mpa_dp_handoff_retries") else 0

    @staticmethod
    def device_bytes():
        """bytes of HBM held through the library: resident index + every pool of every context (mpa_device_bytes)"""
        return int(lib().mpa_device_bytes()) if hasattr(lib(), "mpa_device_bytes") else 0

    @staticmethod
    def stage_clocks(reset=False):
        """{stage: (wall ms, calls)} of mpa_map_batches' stages since the last reset (mpa_stage_clocks)"""
        ms, n = (C.c_double * 5)(), (C.c_int64 * 5)()
        if hasattr(lib(), "mpa_stage_clocks"):
            lib().mpa_stage_clocks(ms, n, 1 if reset else 0)
        return {k: (ms[i], int(n[i])) for i, k in enumerate(("seeding", "planning", "dp", "output", "sketch"))}

    @staticmethod
    def pool_growths():
        """how often a device pool had to be re-allocated so far (mpa_pool_growths)"""
        return int(lib().mpa_pool_growths()) if hasattr(lib(), "mpa_pool_growths") else 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    def __init__(self, handle):
        if not handle:
            raise MpaError("cannot open the index: " + last_error())
        self.h = handle

    @classmethod
    def restore(cls, path):
        return cls(lib().mpa_idx_restore(os.fsencode(path)))

    @classmethod
    def from_nt4(cls, contigs, names=None):
        """contigs: list of uint8 arrays of nt4 codes."""
        n = len(contigs)
        names = names or ["ctg%d" % i for i in range(n)]
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        lens = np.array([len(c) for c in contigs], dtype=np.int64)
        cat = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint8) for c in contigs]))
        return cls(lib().mpa_idx_from_nt4(n, arr, lens.ctypes.data, cat.ctypes.data))

    @classmethod
    def from_fasta(cls, path, n_threads=4):
        """Read a (possibly gzipped) FASTA, pack it and build the k-mer table on the host (index.c:97-136)."""
        import gzip
        op = gzip.open if path.endswith(".gz") else open
        names, parts, cur = [], [], []
        lut = np.full(256, 4, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = lut[ch | 0x20] = i
        with op(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if names:
                        parts.append(np.concatenate(cur) if cur else np.zeros(0, np.uint8))
                    names.append(line[1:].split()[0].decode())
                    cur = []
                else:
                    cur.append(lut[np.frombuffer(line.rstrip(), dtype=np.uint8)])
        if names:
            parts.append(np.concatenate(cur) if cur else np.zeros(0, np.uint8))
        idx = cls.from_nt4(parts, names)
        _check(lib().mpa_idx_build_kmers(idx.h, n_threads))
        return idx

    @classmethod
    def read_fasta(cls, path, idxopt=None):
        """Genome-only index of a FASTA file (mpa_idx_read_fasta); idxopt: (bbit, min_aa_len, kmer, mod_bit, trans_code) or None"""
        io = (C.c_int32 * 5)()
        lib().mpa_idxopt_init(io)
        if idxopt is not None:
            for k, v in enumerate(idxopt):
                io[k] = v
        return cls(lib().mpa_idx_read_fasta(os.fsencode(path), io))

    def build_kmers(self, n_threads=4, ctx=None):
        """The k-mer table (mp_idx_build): on the GPU when a context is given and the device path takes the job, else on the host.
        Returns "gpu" or "host"."""
        if ctx is not None:
            rc = lib().mpa_idx_build_kmers_device(ctx.h, self.h)
            if rc == 0:
                return "gpu"
            if rc != -3:                                      # MPA_ERR_UNSUPPORTED: parameters / memory -> host
                _check(rc)
        _check(lib().mpa_idx_build_kmers(self.h, n_threads))
        return "host"

    def to_device(self, ctx):
        _check(lib().mpa_idx_to_device(ctx.h, self.h))

    def n_ctg(self):
        return lib().mpa_idx_n_ctg(self.h)

    def ctg_len(self, cid):
        return lib().mpa_idx_ctg_len(self.h, cid)

    def ctg_name(self, cid):
        return lib().mpa_idx_ctg_name(self.h, cid).decode()

    def genome_len(self):
        return lib().mpa_idx_genome_len(self.h)

    has_spsc = False

    def set_spsc(self, path, mo, keep_io=False):
        """--spsc: load a splice-score file (mp_set_spsc); adjusts mo.io / mo.io_end unless keep_io.  Before to_device()."""
        n = lib().mpa_idx_set_spsc(self.h, path.encode(), C.byref(mo), 1 if keep_io else 0)
        if n < 0:
            raise MpaError(last_error())
        self.has_spsc = True
        return n

    def get_spsc(self, vid, st, en):
        """the ss[] bytes the reference hands the DP for the window [st,en) of vid, or None without a track"""
        out = np.zeros(max(en - st, 1), dtype=np.uint8)
        n = lib().mpa_idx_get_spsc(self.h, vid, st, en, out.ctypes.data)
        return None if n < 0 else out[:n]

    def get_nt(self, vid, st, en):
        buf = np.zeros(max(en - st, 0), dtype=np.uint8)
        n = lib().mpa_idx_get_nt(self.h, vid, st, en, buf.ctypes.data)
        return buf[:max(n, 0)]

    def dump(self, path):
        _check(lib().mpa_idx_dump(os.fsencode(path), self.h))

    def close(self):
        if self.h:
            lib().mpa_idx_destroy(self.h)
            self.h = None


class Queries:
    """A batch of protein sequences as the C ABI wants them (one concatenated buffer + offsets)."""

    def __init__(self, seqs, names=None):
        self.seqs = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        self.names = names or ["q%d" % i for i in range(len(self.seqs))]
        self.buf = b"".join(self.seqs)
        self.off = np.zeros(len(self.seqs) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in self.seqs], out=self.off[1:])
        self.c = QBatch(len(self.seqs), self.buf, self.off.ctypes.data_as(C.POINTER(C.c_int64)))


def default_mapopt():
    mo = MapOpt()
    lib().mpa_mapopt_init(C.byref(mo))
    return mo


def dpopt_from(mo):
    dp = DpOpt()
    lib().mpa_dpopt_from_mapopt(C.byref(mo), C.byref(dp))
    return dp


class Result:
    """Owner of an mpa_result_t (the structured hits of one batch)."""

    def __init__(self, handle):
        self.h = handle

    def n_hit(self):
        return lib().mpa_result_n_hit(self.h)

    def n_output(self, mo, queries):
        """ids the output of this batch consumes (mpa_result_n_output)"""
        return lib().mpa_result_n_output(C.byref(mo), C.byref(queries.c), self.h)

    def close(self):
        if self.h:
            lib().mpa_result_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_output(idx, mo, queries, result, id0=0):
    """mpa_format_output(): the text of one batch whose first printed hit gets id0 + 1.  Returns (text, id after the batch)."""
    names = (C.c_char_p * len(queries.names))(*[n.encode() for n in queries.names])
    out = C.c_void_p()
    idc = C.c_int64(id0)
    n = lib().mpa_format_output(idx.h, C.byref(mo), C.byref(queries.c), names, result.h, C.byref(idc), C.byref(out))
    if n < 0:
        raise MpaError(last_error())
    txt = C.string_at(out.value, n)
    lib().mpa_free(out)
    return txt, idc.value


def map_batch(ctx, idx, mo, queries, n_threads=1):
    """mpa_map_batch(): one blocking batch; returns its Result."""
    res = C.c_void_p()
    _check(lib().mpa_map_batch(ctx.h, idx.h, C.byref(mo), C.byref(queries.c), n_threads, C.byref(res)))
    return Result(res.value)


CLAIM_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p)


def map_batches(ctx, idx, mo, batches, n_threads=1, keep_results=False, want_text=True, claim=None):
    """mpa_map_batches(): a stream of Queries batches through the pipelined mapper.  Returns the list of output texts
    (bytes, one per batch; the hit-id counter runs across the batches as in one output file); with keep_results the pair
    (texts, [Result]); with want_text=False no text is produced (texts are None).
    claim: mpa_map_batches_claim() -- `batches` is a whole job shared with other processes and claim() returns the index of
    the next batch this process should map (-1: none left); the return value is then (order, texts[, results]) for the
    batches this call mapped, in the order it claimed them."""
    n = len(batches)
    if n == 0:
        return (([], [], []) if keep_results else ([], [])) if claim else (([], []) if keep_results else [])
    qb = (QBatch * n)(*[b.c for b in batches])
    res = (C.c_void_p * n)()
    idc = C.c_int64(0)
    names = text = tlen = None
    if want_text:
        name_arrays = [(C.c_char_p * len(b.names))(*[x.encode() for x in b.names]) for b in batches]
        names = (C.POINTER(C.c_char_p) * n)(*[C.cast(a, C.POINTER(C.c_char_p)) for a in name_arrays])
        text = (C.c_void_p * n)()
        tlen = (C.c_int64 * n)()
    order = None
    if claim is not None:
        err = []
        def _claim(_user):
            try:
                return int(claim())
            except BaseException as e:                          # (an exception must not unwind through the C pipeline)
                err.append(e)
                return -1
        cb = CLAIM_FN(_claim)
        n_mapped = C.c_int32(0)
        order_c = (C.c_int32 * n)()
        f = lib().mpa_map_batches_claim
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      CLAIM_FN, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(ctx.h, idx.h, C.byref(mo), n, qb, names, n_threads, res, C.byref(idc) if want_text else None, text, tlen, cb, None, C.byref(n_mapped), order_c))
        if err:
            raise err[0]
        n = n_mapped.value
        order = [int(order_c[k]) for k in range(n)]
    elif want_text:
        _check(lib().mpa_map_batches(ctx.h, idx.h, C.byref(mo), n, qb, names, n_threads, res, C.byref(idc), text, tlen))
    else:
        _check(lib().mpa_map_batches(ctx.h, idx.h, C.byref(mo), n, qb, None, n_threads, res, None, None, None))
    out, results = [], []
    for k in range(n):
        if want_text:
            out.append(C.string_at(text[k], tlen[k]))
            lib().mpa_free(C.c_void_p(text[k]))
        else:
            out.append(None)
        if keep_results:
            results.append(Result(res[k]))
        else:
            lib().mpa_result_destroy(C.c_void_p(res[k]))
    if order is not None:
        return (order, out, results) if keep_results else (order, out)
    return (out, results) if keep_results else out


def map_batches_multi(ctxs, idx, mo, batches, n_threads=1):
    """mpa_map_batches_multi(): one job over several device contexts of THIS process (one pipeline per context, batches claimed
    from a shared counter).  Returns the output texts in input order -- the bytes a single pipeline produces."""
    n = len(batches)
    if n == 0:
        return []
    qb = (QBatch * n)(*[b.c for b in batches])
    res = (C.c_void_p * n)()
    idc = C.c_int64(0)
    name_arrays = [(C.c_char_p * len(b.names))(*[x.encode() for x in b.names]) for b in batches]
    names = (C.POINTER(C.c_char_p) * n)(*[C.cast(a, C.POINTER(C.c_char_p)) for a in name_arrays])
    text = (C.c_void_p * n)()
    tlen = (C.c_int64 * n)()
    hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    f = lib().mpa_map_batches_multi
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(len(ctxs), hs, idx.h, C.byref(mo), n, qb, names, n_threads, res, C.byref(idc), text, tlen))
    out = []
    for k in range(n):
        out.append(C.string_at(text[k], tlen[k]))
        lib().mpa_free(C.c_void_p(text[k]))
        lib().mpa_result_destroy(C.c_void_p(res[k]))
    return out


def dp_run(ctx, idx, dpopt, queries, tasks, wide32=False):
    """Run a batch of DP calls on the GPU.  tasks: numpy array of DP_TASK.  Returns (results, cigar_pool).
    wide32: the 32-bit operator (mpa_dp_run32 = ns_global_gs32b per call) instead of the int16 one."""
    tasks = np.ascontiguousarray(tasks, dtype=DP_TASK)
    rst = np.zeros(len(tasks), dtype=DP_RST)
    pool = C.POINTER(C.c_uint32)()
    n_pool = C.c_int64(0)
    fn = lib().mpa_dp_run32 if wide32 else lib().mpa_dp_run
    fn.argtypes = lib().mpa_dp_run.argtypes
    _check(fn(ctx.h, idx.h, C.byref(dpopt), C.byref(queries.c), len(tasks), tasks.ctypes.data,
              rst.ctypes.data, C.byref(pool), C.byref(n_pool)))
    cig = np.ctypeslib.as_array(pool, (max(n_pool.value, 1),))[:n_pool.value].copy() if n_pool.value else np.zeros(0, np.uint32)
    lib().mpa_free(pool)
    return rst, cig
