"""The N>1 path on CPU: two gloo ranks map disjoint protein shards (host logic + oracle DP executor) and
gather the PAF text to rank 0; the concatenation must equal the single-process output."""
import os
import subprocess
import sys
import numpy as np
import miniprot_amd.dist as mdist
import refbind

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import torch, torch.distributed as dist
import miniprot_amd as mpa, miniprot_amd.dist as mdist, golden
from hostpipe import map_batch, oracle_executor
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
case = golden.SYNTH_CASES[0]
contigs, prots, names = golden.synth_inputs(case)
prots, names = prots[:10], names[:10]
idx = mpa.Index.from_nt4(contigs, ["chr1"])
mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
lo, hi = mdist.shard_range(len(prots), rank, world)
paf = map_batch(idx, golden.mapopt_for(case), mpa.Queries(prots[lo:hi], names[lo:hi]), oracle_executor)
dist.barrier()
parts = mdist.gather_bytes(paf, rank, world, torch.device("cpu"))
if rank == 0:
    open(sys.argv[2], "wb").write(b"".join(parts))
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            cuts = [mdist.shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))


def test_two_rank_gloo_gather_equals_single_process(tmp_path, oracle_built):
    import miniprot_amd as mpa
    import golden
    from hostpipe import map_batch, oracle_executor
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "gathered.paf"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", str(script), refbind.ROOT, str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    case = golden.SYNTH_CASES[0]
    contigs, prots, names = golden.synth_inputs(case)
    idx = mpa.Index.from_nt4(contigs, ["chr1"])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
    single = map_batch(idx, golden.mapopt_for(case), mpa.Queries(prots[:10], names[:10]), oracle_executor)
    assert out.read_bytes() == single
