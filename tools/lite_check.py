import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, miniprot_amd as mpa, refbind
from dpgen import make_task
from dputil import build_workload, oracle_eval, dpopt_from_params, compare
rng = np.random.default_rng(5)
P = refbind.DpParams(refbind.mapping_matrix(23))
pairs = [make_task(rng, max_intron=3000, flank=300, al=int(rng.integers(1, 129))) for _ in range(300)]
contigs, queries, tasks, meta = build_workload(pairs, rng, modes=("cigar",), io=P.io)
ctx = mpa.Context(0); idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)
rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
st = ctx.dp_stats()
bad, msg = compare(rst, cig, oracle_eval(pairs, meta, P), meta, pairs)
print("calls", len(tasks), "bad", len(bad), "n_glob", st["n_glob"], "n_ckpt", st["n_ckpt"], "cells_glob", st["cells_glob"], "cells_ckpt", st["cells_ckpt"], "walk_blocks", st["walk_blocks"], "rows of ckpt calls", sum(t["nl"] for t in tasks if t["nl"] >= 384))
print(msg[:3000])
