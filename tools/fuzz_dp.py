"""Randomised parity of mpa_dp_run() against the oracle: python tools/fuzz_dp.py [seconds] [first seed].  Every call class of the
round kernel (asm extension rows, the 128-column one-call-per-wave classes, the checkpointed traceback at several row thresholds;
third argument "wide": the multi-wave extension classes of 129..1024 columns)."""
import sys, os, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, miniprot_amd as mpa, refbind
from dpgen import make_task
from dputil import build_workload, oracle_eval, dpopt_from_params, compare
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"     # the multi-wave extension classes: 129..1024 columns, pairs of calls, split groups
t0 = time.time()
n_calls = n_bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    os.environ["MPA_DP_LITE_MIN"] = str(int(rng.choice([3, 50, 100, 384, 384, 1000])))
    fs = int(rng.choice([23, 23, 17, 10]))
    P = refbind.DpParams(refbind.mapping_matrix(fs), go=int(rng.choice([11, 11, 5, 13])), ge=int(rng.choice([1, 1, 2, 3])), io=int(rng.choice([29, 29, 40, 19])), fs=fs,
                         xdrop=int(rng.choice([100, 100, 30, 300])), end_bonus=int(rng.choice([5, 0, 11])), sp=tuple(rng.choice([0, 1], 1)) and ((8, 15, 21, 30, 0, 0) if rng.random() < 0.6 else (8, 15, 21, 30, 4, 4)),
                         ie_coef=float(rng.choice([0.5, 0.5, 1.0, 0.0])))
    pairs = []
    for _ in range(60 if wide else 250):
        al = int(rng.choice([rng.integers(1, 17), rng.integers(8, 65), rng.integers(60, 130), rng.integers(120, 300)], p=[0.2, 0.4, 0.3, 0.1]))
        if wide:
            al = int(rng.choice([rng.integers(129, 257), rng.integers(257, 513), rng.integers(513, 1025), rng.integers(1, 129)], p=[0.3, 0.3, 0.3, 0.1]))
        pairs.append(make_task(rng, al=al, max_intron=int(rng.choice([200, 2000, 12000])), flank=int(rng.choice([5, 100, 800, 4000])), p_intron=float(rng.choice([0.03, 0.1, 0.2])),
                               p_fs=float(rng.choice([0.0, 0.01, 0.05])), p_indel=float(rng.choice([0.0, 0.02, 0.06])), p_n=float(rng.choice([0.0, 0.002, 0.01]))))
    contigs, queries, tasks, meta = build_workload(pairs, rng, io=P.io)
    ctx = mpa.Context(0); idx = mpa.Index.from_nt4(contigs); idx.to_device(ctx)
    rst, cig = mpa.dp_run(ctx, idx, dpopt_from_params(P), queries, tasks)
    st = ctx.dp_stats()
    bad, msg = compare(rst, cig, oracle_eval(pairs, meta, P), meta, pairs)
    idx.close(); ctx.close()
    n_calls += len(tasks); n_bad += len(bad)
    print("seed %d lite_min %s go %d ge %d io %d fs %d xdrop %d: %d calls (%d checkpointed, %d blocks recomputed), %d differ" % (seed, os.environ["MPA_DP_LITE_MIN"], P.go, P.ge, P.io, P.fs, P.xdrop, len(tasks), st["n_ckpt"], st["walk_blocks"], len(bad)), flush=True)
    if bad:
        print(msg[:2000])
    seed += 1
print("fuzz: %d calls, %d differ from the oracle, %.0f s" % (n_calls, n_bad, time.time() - t0))
sys.exit(1 if n_bad else 0)
