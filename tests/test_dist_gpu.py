"""The N>1 path with the HIP mapper: two ranks (gloo rendezvous, both on GPU 0 -- the box has one) share the mini-batches of one
job through the dynamic deal -- every rank's mpa_map_batches_claim() stream pulls the next unclaimed batch from the shared counter
(miniprot_amd/dist.py) -- and rank 0 must end up with the bytes of the single-process run: PAF, and GFF3 whose hit ids run across
the whole output file (map.c:306) and therefore across the ranks.  (bench.py --gpus N drives exactly this with RCCL.)"""
import os
import subprocess
import sys
import pytest
import refbind

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import torch, torch.distributed as dist
import miniprot_amd as mpa, miniprot_amd.dist as mdist, golden
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
case = golden.SYNTH_CASES[1]
contigs, prots, names = golden.synth_inputs(case)
idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
ctx = mpa.Context(0)
idx.to_device(ctx)
n = len(prots)
cuts = [0, n // 7, n // 7, n // 3, n // 2, n - 3, n]               # six uneven mini-batches, one of them empty
batches = [mpa.Queries(prots[a:b], names[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
for tag, flag in (("paf", 0), ("gff", 0x8)):
    mo = golden.mapopt_for(case)
    mo.flag |= flag
    needs_ids = bool(flag)
    def map_claimed(claim):
        if needs_ids:
            order, _, results = mpa.map_batches(ctx, idx, mo, batches, 2, keep_results=True, want_text=False, claim=claim)
            return order, results
        return mpa.map_batches(ctx, idx, mo, batches, 2, claim=claim)
    texts, n_ids = mdist.run_sharded(len(batches), rank, world, torch.device("cpu"), needs_ids, None,
                                     lambda k, r: r.n_output(mo, batches[k]), lambda k, r, id0: mpa.format_output(idx, mo, batches[k], r, id0)[0],
                                     map_claimed=map_claimed)
    if rank == 0:
        open(sys.argv[2] + "." + tag, "wb").write(b"".join(texts))
    open(sys.argv[2] + ".rank%d.%s.n" % (rank, tag), "w").write("ok")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_share_a_job_through_the_hip_mapper(tmp_path):
    import miniprot_amd as mpa
    import golden
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "gathered")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", str(script), refbind.ROOT, out], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    case = golden.SYNTH_CASES[1]
    contigs, prots, names = golden.synth_inputs(case)
    idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
    ctx = mpa.Context(0)
    idx.to_device(ctx)
    for tag, flag in (("paf", 0), ("gff", 0x8)):
        mo = golden.mapopt_for(case)
        mo.flag |= flag
        single = b"".join(mpa.map_batches(ctx, idx, mo, [mpa.Queries(prots, names)], 2))
        got = open(out + "." + tag, "rb").read()
        assert got == single, tag
        if tag == "paf":                                      # ... which are the reference's bytes
            assert golden.file_header(case) + got == open(golden.path(case["name"] + ".ref.paf"), "rb").read()
    idx.close()
    ctx.close()
