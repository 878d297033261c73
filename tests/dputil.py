"""Helpers shared by the operator-level DP tests: lay random (window, protein) pairs out as a genome +
query batch + mpa_dp_task_t array, and evaluate the same calls with the oracle."""
import numpy as np
import miniprot_amd as mpa
import refbind
from dpgen import make_task

COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def build_workload(pairs, rng, modes=("cigar", "left", "right"), io=29, io_alt=19, n_ctg=3):
    """pairs: list of (nt bytes, aa bytes).  Every pair is placed on a random contig and strand; for each
    requested mode one task is emitted.  Returns (contigs, queries, tasks, meta) where meta[k] =
    (pair index, flag, io)."""
    ctgs = [[] for _ in range(n_ctg)]
    pos = [0] * n_ctg
    place = []
    for nt, aa in pairs:
        c = int(rng.integers(0, n_ctg))
        rev = int(rng.integers(0, 2))
        arr = np.frombuffer(nt, dtype=np.uint8)
        gap = rng.integers(0, 4, int(rng.integers(0, 30))).astype(np.uint8)
        ctgs[c].append(gap)
        pos[c] += len(gap)
        fwd = COMP[arr[::-1]] if rev else arr
        ctgs[c].append(fwd)
        place.append((c, rev, pos[c], len(arr)))
        pos[c] += len(arr)
    contigs = [np.concatenate(x) if x else np.zeros(1, np.uint8) for x in ctgs]
    lens = [len(x) for x in contigs]
    tasks = []
    meta = []
    flags = {"cigar": mpa.F_CIGAR, "left": mpa.F_EXT_LEFT, "right": mpa.F_EXT_RIGHT}
    for k, ((nt, aa), (c, rev, p, nl)) in enumerate(zip(pairs, place)):
        nt_off = lens[c] - (p + nl) if rev else p
        for m in modes:
            fl = flags[m]
            this_io = io if (fl == mpa.F_CIGAR or rng.random() < 0.7) else io_alt
            tasks.append((nt_off, c << 1 | rev, nl, k, 0, len(aa), fl, this_io, len(tasks)))
            meta.append((k, fl, this_io))
    tasks = np.array(tasks, dtype=mpa.DP_TASK)
    queries = mpa.Queries([aa for _, aa in pairs])
    return contigs, queries, tasks, meta


def oracle_eval(pairs, meta, P):
    out = []
    for k, fl, io in meta:
        nt, aa = pairs[k]
        PP = refbind.DpParams(P.mat, go=P.go, ge=P.ge, io=io, fs=P.fs, xdrop=P.xdrop, end_bonus=P.end_bonus, sp=P.sp,
                              sp_null_bonus=P.sp_null_bonus, ie_coef=P.ie_coef)
        out.append(refbind.ora_nasw(nt, aa, PP, fl))
    return out


def dpopt_from_params(P):
    dp = mpa.DpOpt()
    dp.go, dp.ge, dp.fs, dp.xdrop, dp.end_bonus, dp.ie_coef = P.go, P.ge, P.fs, P.xdrop, P.end_bonus, P.ie_coef
    for i in range(6):
        dp.sp[i] = P.sp[i]
    flat = np.ascontiguousarray(P.mat, dtype=np.int8).reshape(-1)
    for i in range(484):
        dp.mat[i] = int(flat[i])
    return dp


def compare(rst, cig, expect, meta, pairs, max_report=5):
    bad = []
    for k, (r, e) in enumerate(zip(rst, expect)):
        got_c = [int(x) for x in cig[r["cigar_off"]: r["cigar_off"] + r["n_cigar"]]]
        got = (int(r["nt_len"]), int(r["aa_len"]), int(r["score"]), got_c)
        if got != e:
            bad.append((k, got, e))
    msgs = []
    for k, got, e in bad[:max_report]:
        pk, fl, io = meta[k]
        msgs.append("task %d flag %d io %d nl %d al %d\n  gpu %s %s\n  ora %s %s" % (
            k, fl, io, len(pairs[pk][0]), len(pairs[pk][1]), got[:3], refbind.cigar_str(got[3]), e[:3], refbind.cigar_str(e[3])))
    return bad, "\n".join(msgs)
