"""Drive the host stage machine (mpa_batch_*) with a pluggable DP executor.

`oracle_executor` evaluates every DP task with the C restatement in oracle/ -- this is how the CPU-only
tests check the HOST logic of the product against the reference's output.  The product's own path
(mpa_map_batch) uses the HIP executor instead and never touches oracle/."""
import ctypes as C
import gzip
import numpy as np
import miniprot_amd as mpa
import refbind


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    names, seqs, cur = [], [], []
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if names:
                    seqs.append(b"".join(cur))
                names.append(line[1:].split()[0].decode())
                cur = []
            elif line:
                cur.append(line)
    if names:
        seqs.append(b"".join(cur))
    return names, seqs


def oracle_executor(idx, queries, dpopt, tasks):
    mat = np.array(list(dpopt.mat), dtype=np.int8).reshape(22, 22)
    rst = np.zeros(len(tasks), dtype=mpa.DP_RST)
    pool = []
    for k, t in enumerate(tasks):
        nt = idx.get_nt(int(t["vid"]), int(t["nt_off"]), int(t["nt_off"]) + int(t["nl"]))
        aa = queries.seqs[int(t["qid"])][int(t["aa_off"]): int(t["aa_off"]) + int(t["al"])]
        P = refbind.DpParams(mat, go=dpopt.go, ge=dpopt.ge, io=int(t["io"]), fs=dpopt.fs, xdrop=dpopt.xdrop,
                             end_bonus=dpopt.end_bonus, sp=tuple(dpopt.sp), ie_coef=dpopt.ie_coef)
        ss = None
        if idx.has_spsc:
            # the reference cuts ss[] once per alignment window and hands sub-windows to the DP (align.c:270-327): every position
            # but the window's first carries its score; MPA_F_SS_SKIP0 marks the calls that start at that first position
            st = int(t["nt_off"])
            ss = idx.get_spsc(int(t["vid"]), st - 1, st + int(t["nl"]))[1:] if st > 0 else idx.get_spsc(int(t["vid"]), 0, int(t["nl"]))
            ss = ss.copy()
            if int(t["flag"]) & 8:
                ss[0] = 0xff
            P.sp_null_bonus = dpopt.sp_null_bonus
        nt_len, aa_len, score, cig = refbind.ora_nasw(bytes(nt), aa, P, int(t["flag"]) & 7, ss)
        rst[k] = (nt_len, aa_len, score, len(cig), len(pool))
        pool.extend(cig)
    return rst, np.array(pool if pool else [0], dtype=np.uint32)


def gpu_executor(ctx):
    def run(idx, queries, dpopt, tasks):
        return mpa.dp_run(ctx, idx, dpopt, queries, tasks)
    return run


def map_batch_result(idx, mo, queries, executor, n_threads=1):
    """The batch through the stage machine with `executor` for the DP rounds; returns the structured result (mpa.Result)."""
    L = mpa.lib()
    b = L.mpa_batch_begin(idx.h, C.byref(mo), C.byref(queries.c), n_threads)
    if not b:
        raise mpa.MpaError(mpa.last_error())
    dpopt = mpa.DpOpt()
    while True:
        ptr = C.c_void_p()
        n = L.mpa_batch_dp_tasks(b, C.byref(ptr), C.byref(dpopt))
        if n <= 0:
            break
        tasks = np.frombuffer((C.c_char * (n * mpa.DP_TASK.itemsize)).from_address(ptr.value), dtype=mpa.DP_TASK).copy()
        rst, pool = executor(idx, queries, dpopt, tasks)
        rst = np.ascontiguousarray(rst)
        pool = np.ascontiguousarray(pool, dtype=np.uint32)
        mpa._check(L.mpa_batch_dp_results(b, rst.ctypes.data, pool.ctypes.data))
    return mpa.Result(L.mpa_batch_finish(b))


def map_batch(idx, mo, queries, executor, n_threads=1):
    """Returns the output text (bytes) of the batch."""
    res = map_batch_result(idx, mo, queries, executor, n_threads)
    return mpa.format_output(idx, mo, queries, res)[0]


def map_batch_gpu(ctx, idx, mo, queries, n_threads=1):
    res = mpa.map_batch(ctx, idx, mo, queries, n_threads)
    return mpa.format_output(idx, mo, queries, res)[0]
