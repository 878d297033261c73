"""The N>1 path on CPU: two gloo ranks share the mini-batches of one job (host logic + oracle DP executor, the same
miniprot_amd.dist.run_sharded() that bench.py --gpus N drives with the HIP mapper) and rank 0 must end up with the bytes of
the single-process run -- for PAF, and for GFF3/GTF whose hit ids run across the whole output file (map.c:306)."""
import os
import subprocess
import sys
import numpy as np
import pytest
import miniprot_amd.dist as mdist
import refbind

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import torch, torch.distributed as dist
import miniprot_amd as mpa, miniprot_amd.dist as mdist, golden
from hostpipe import map_batch_result, oracle_executor
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
case = golden.SYNTH_CASES[0]
contigs, prots, names = golden.synth_inputs(case)
idx = mpa.Index.from_nt4(contigs, ["chr1"])
mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
cuts = [0, 3, 4, 9, 9, 14]                                  # five uneven mini-batches, one of them empty
batches = [mpa.Queries(prots[a:b], names[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
for tag, flag in (("paf", 0x4), ("gff", 0x4 | 0x8), ("gtf", 0x4 | 0x20)):
    mo = golden.mapopt_for(case)
    mo.flag |= flag
    needs_ids = bool(flag & (0x8 | 0x20))
    def map_mine(ks):
        rs = [map_batch_result(idx, mo, batches[k], oracle_executor) for k in ks]
        return rs if needs_ids else [mpa.format_output(idx, mo, batches[k], r)[0] for k, r in zip(ks, rs)]
    texts, n_ids = mdist.run_sharded(len(batches), rank, world, torch.device("cpu"), needs_ids, map_mine,
                                     lambda k, r: r.n_output(mo, batches[k]), lambda k, r, id0: mpa.format_output(idx, mo, batches[k], r, id0)[0])
    if rank == 0:
        open(sys.argv[2] + "." + tag, "wb").write(b"".join(texts))
    # the dynamic deal: both ranks pull batch indices from the shared counter until the job is exhausted (whoever is faster maps
    # more); the gathered output does not depend on who mapped what
    def map_claimed(claim):
        ks, rs = [], []
        while True:
            k = claim()
            if k < 0:
                break
            ks.append(k)
            rs += map_mine([k])
        return ks, rs
    texts, n_ids = mdist.run_sharded(len(batches), rank, world, torch.device("cpu"), needs_ids, None,
                                     lambda k, r: r.n_output(mo, batches[k]), lambda k, r, id0: mpa.format_output(idx, mo, batches[k], r, id0)[0],
                                     map_claimed=map_claimed)
    if rank == 0:
        open(sys.argv[2] + "." + tag + ".dyn", "wb").write(b"".join(texts))
# a rank whose mapping fails (here: rank 1's claim callback raises) must not leave the others waiting in the gather: every rank raises
def map_broken(claim):
    if rank == 1:
        raise ValueError("boom")
    ks = []
    while claim() >= 0:
        ks.append(0)
    return [], []
try:
    mdist.run_sharded(len(batches), rank, world, torch.device("cpu"), False, None, None, None, map_claimed=map_broken)
    raised = False
except RuntimeError as e:
    raised = "mapping failed" in str(e)
open(sys.argv[2] + ".raised.%d" % rank, "w").write("1" if raised else "0")
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            cuts = [mdist.shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert sorted(k for r in range(w) for k in mdist.my_batches(n, r, w)) == list(range(n))


def test_two_rank_gloo_job_equals_single_process(tmp_path, oracle_built):
    import miniprot_amd as mpa
    import golden
    from hostpipe import map_batch_result, oracle_executor
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "gathered")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", str(script), refbind.ROOT, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out + ".raised.0").read() == "1" and open(out + ".raised.1").read() == "1", "a failure on one rank did not reach every rank"
    case = golden.SYNTH_CASES[0]
    contigs, prots, names = golden.synth_inputs(case)
    idx = mpa.Index.from_nt4(contigs, ["chr1"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
    q = mpa.Queries(prots[:14], names[:14])
    for tag, flag in (("paf", 0x4), ("gff", 0x4 | 0x8), ("gtf", 0x4 | 0x20)):
        mo = golden.mapopt_for(case)
        mo.flag |= flag
        single = mpa.format_output(idx, mo, q, map_batch_result(idx, mo, q, oracle_executor))[0]
        got = open(out + "." + tag, "rb").read()
        assert got == single, tag
        assert open(out + "." + tag + ".dyn", "rb").read() == single, tag + " (dynamic deal)"
        if tag == "gff":                                        # the ids really do run across the ranks' batches
            ids = [l.split(b"ID=MP")[1][:6] for l in got.split(b"\n") if b"\tmRNA\t" in l]
            assert len(ids) > 5 and ids == [b"%06d" % (k + 1) for k in range(len(ids))]
