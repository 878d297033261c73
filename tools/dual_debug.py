import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, miniprot_amd as mpa, refbind
from dpgen import make_task
from dputil import build_workload, oracle_eval, dpopt_from_params, compare
P = refbind.DpParams(refbind.mapping_matrix(23))
for al, kw in ((100, dict(p_intron=0.0, flank=3)), (70, dict(p_intron=0.0, flank=3)), (100, dict(p_intron=0.0, flank=3, p_sub=0, p_indel=0, p_fs=0, p_n=0)), (100, dict(max_intron=300, flank=40)), (128, dict(max_intron=2000, flank=40))):
    rng = np.random.default_rng(7)
    pairs = [make_task(rng, al=al, **kw) for _ in range(6)]
    contigs, queries, tasks, meta = build_workload(pairs, rng, modes=("cigar",), io=P.io)
    ctx = mpa.Context(0); idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)
    rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
    exp = oracle_eval(pairs, meta, P)
    st = ctx.dp_stats()
    print("al", al, kw, "n_ckpt", st["n_ckpt"], "blocks", st["walk_blocks"])
    for k in range(len(tasks)):
        print("   nl %5d al %3d gpu score %6d oracle %6d  ncig %d/%d" % (tasks[k]["nl"], tasks[k]["al"], rst[k]["score"], exp[k][2], rst[k]["n_cigar"], len(exp[k][3])))
    idx.close(); ctx.close()
