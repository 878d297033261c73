import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, miniprot_amd as mpa, refbind
from dpgen import make_task
from dputil import *
ctx = mpa.Context(0)
P = refbind.DpParams(refbind.mapping_matrix(23))
def run(pairs, modes, seed=1):
    rng = np.random.default_rng(seed)
    contigs, queries, tasks, meta = build_workload(pairs, rng, modes=modes, io=P.io, io_alt=P.io)
    idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)
    rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
    exp = oracle_eval(pairs, meta, P)
    bad, msg = compare(rst, cig, exp, meta, pairs, max_report=2)
    idx.close()
    return len(bad), len(tasks), msg
rng = np.random.default_rng(5)
for al in (10, 30, 50):
    nb = 0
    for trial in range(6):
        p = [make_task(rng, al=al, p_indel=0.0, flank=40)]
        b, n, msg = run(p, ("right",))
        nb += b
        if b: print(msg.replace("\n", " | ")[:200])
    print("single right al", al, "bad", nb, "/ 6")
p = [make_task(rng, al=10, p_indel=0.0, flank=40) for _ in range(2)]
print("two tasks (lo/hi of group0?)", run(p, ("right",))[:2])
p = [make_task(rng, al=10, p_indel=0.0, flank=40) for _ in range(8)]
r = run(p, ("right",)); print("eight", r[:2], r[2].replace("\n"," | ")[:300])
p = [make_task(rng, al=10, p_indel=0.0, flank=400) for _ in range(1)]
r = run(p, ("right",)); print("one long flank", r[:2], r[2].replace("\n"," | ")[:300])
