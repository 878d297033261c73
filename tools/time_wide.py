"""Kernel-time microbenchmark of the DP classes: n_tasks right-extension calls (flag 4) or traceback calls (flag 1) of `al` columns x
`nl` rows (results are not checked here).  python tools/time_wide.py al nl n_tasks [flag] [antidiag]
antidiag = 1: the 32-column extension class on the anti-diagonal prototype (dp_antidiag.hip) instead of the row sweep."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, miniprot_amd as mpa
al, nl, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flag = int(sys.argv[4]) if len(sys.argv) > 4 else 4      # 4 = right extension (score only), 1 = global with traceback
rng = np.random.default_rng(1)
g = rng.integers(0, 4, nl * n + 1000).astype(np.uint8)
idx = mpa.Index.from_nt4([g], ["c"])
ctx = mpa.Context(0); idx.to_device(ctx)
antidiag = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if antidiag:
    import ctypes as C
    mpa.lib().mpa_dbg_antidiag.argtypes = [C.c_void_p, C.c_int]
    mpa.lib().mpa_dbg_antidiag(ctx.h, 1)
aa = bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), al * n).tolist())
q = mpa.Queries([aa[i * al:(i + 1) * al] for i in range(n)])
tasks = np.zeros(n, mpa.DP_TASK)
for i in range(n):
    tasks[i]["nt_off"] = i * nl; tasks[i]["vid"] = 0; tasks[i]["nl"] = nl; tasks[i]["qid"] = i; tasks[i]["aa_off"] = 0; tasks[i]["al"] = al
    tasks[i]["flag"] = flag; tasks[i]["io"] = 29
dp = mpa.dpopt_from(mpa.default_mapopt())
dp.xdrop = 32000                                   # never stop: time all rows
for it in range(3):
    mpa.dp_run(ctx, idx, dp, q, tasks)
    st = ctx.dp_stats()
    ms = st["ms_ext"] if flag != 1 else st["ms_glob"]
    tot = ctx.dp_stats(total=True, reset=True)
    if tot["ms_round"] > 0:                        # worker pool (default): the round is a k_dp_worker launch on the worker stream
        ms = tot["ms_round"]
    print("al %d nl %d x %d flag %d%s: %.2f ms -> %.1f ns/row (backtrack %.2f ms)" % (al, nl, n, flag, " ANTIDIAG" if antidiag else "", ms, ms * 1e6 / nl, st["ms_backtrack"]))
