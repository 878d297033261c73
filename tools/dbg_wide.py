import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, miniprot_amd as mpa, refbind
from dpgen import make_task
from dputil import *
ctx = mpa.Context(0)
P = refbind.DpParams(refbind.mapping_matrix(23))
def run(pairs, modes, seed=1):
    rng = np.random.default_rng(seed)
    contigs, queries, tasks, meta = build_workload(pairs, rng, modes=modes, io=P.io)
    idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)
    rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
    exp = oracle_eval(pairs, meta, P)
    bad, msg = compare(rst, cig, exp, meta, pairs, max_report=3)
    idx.close()
    return len(bad), len(tasks), msg
rng = np.random.default_rng(5)
# 1. a single wide task alone (other half empty)
for al in (70, 100, 128):
    for trial in range(3):
        p = [make_task(rng, al=al, p_indel=0.0, flank=300)]
        for modes in (("right",), ("left",)):
            nb, nt, msg = run(p, modes)
            print("single al", al, modes, "bad", nb, "/", nt, msg.replace("\n", " | ")[:300] if nb else "")
# 2. pairs
p = [make_task(rng, al=100, p_indel=0.0, flank=300) for _ in range(2)]
print("pair right", run(p, ("right",))[:2])
p = [make_task(rng, al=100, p_indel=0.0, flank=50) for _ in range(8)]
print("8 right", run(p, ("right",))[:2])
p = [make_task(rng, al=100, p_indel=0.0, flank=3000) for _ in range(8)]
r = run(p, ("right",)); print("8 right long flank", r[:2], r[2][:600])
