#!/usr/bin/env python3
"""bench.py -- aligned protein residues/sec of the MI355X-native miniprot hot path.

A "step" is one pass of the whole per-query hot path (seed -> chain -> refine -> spliced DP on the GPU ->
CIGAR/statistics -> PAF text) over one batch of synthetic proteins against a synthetic genome whose index
is already resident in HBM.  Workload = BASELINE.json configs[1]: synthetic 50 Mbp genome x 1k ~400 aa
proteins, prebuilt index, flags `-u` (generator: tools/gen_synth.py).

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The K steps are K mini-batches handed to mpa_map_batches() as one stream (the library pipelines host seeding, GPU DP
rounds and formatting across consecutive batches; --no-stream times one blocking mpa_map_batch() call per step).

N>1: proteins are sharded over ranks (one process per GPU, no data-path collective); every rank maps
its own shard of the same size (weak scaling) and the PAF text is gathered to rank 0 over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # the DP executor runs one kernel class per stream; let them overlap
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
# packed-int16 VALU peak: 256 CU x 4 SIMD x 32 lanes/clk x 2 (packed) x 2.4 GHz  (MI355X_MICROARCH.md: SIMD-32, 2400 MHz)
VALU_I16_PEAK = 256 * 4 * 32 * 2 * 2.4e9
OPS_PER_CELL_EXT = 32            # int16 ops per DP cell, score-only loop of the reference


def aligned_residues(paf):
    n = 0
    for line in paf.split(b"\n"):
        f = line.split(b"\t", 6)
        if len(f) > 5 and f[5] != b"*":
            n += int(f[3]) - int(f[2])
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mb", type=float, default=50.0)
    ap.add_argument("--n-prot", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=0, help="host threads for the non-DP stages (0 = all cores / ranks)")
    ap.add_argument("--cpu-sample", type=int, default=200, help="proteins in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-replicas", type=int, default=20, help="the CPU baseline maps the protein set this many times over (one FASTA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="one blocking mpa_map_batch() call per step instead of handing the K steps "
                    "to mpa_map_batches() as a pipelined stream of mini-batches")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if "RANK" in os.environ:                                # launched by torch.distributed.run (any world size)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DP kernels have no CPU fallback")
    torch.cuda.set_device(local_rank)

    import miniprot_amd as mpa
    from miniprot_amd import dist as mdist
    import gen_synth
    from hostpipe import map_batch_gpu

    ncpu = os.cpu_count() or 1
    # host threads for the non-DP stages: share the cores between the ranks; beyond ~128 threads SMT contention loses
    n_threads = a.threads if a.threads > 0 else max(1, min(128, ncpu // max(world, 1)))

    # ---- inputs (untimed): genome, index, proteins.  Every rank builds the same genome; proteins of a
    # rank are that rank's shard (same planted genes, rank-specific divergence), so per-GPU work is fixed.
    t0 = time.time()
    contigs, prots0, names0, planted = gen_synth.generate(int(a.genome_mb * 1e6), 1, a.n_prot, 12, return_planted=True)
    idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, max(2, n_threads)))
    if rank == 0:
        prots, names = prots0, names0
    else:                                                 # same planted genes, rank-specific divergence
        rng = np.random.default_rng(1000 + rank)
        prots, names = [gen_synth.mutate(rng, pr) for pr in planted], ["r%d_%s" % (rank, n) for n in names0]
    q = mpa.Queries(prots, names)
    ctx = mpa.Context(local_rank)
    idx.to_device(ctx)
    mo = mpa.default_mapopt()
    mo.flag |= 0x4                                        # -u: print unmapped queries too
    t_setup = time.time() - t0

    def step():
        return map_batch_gpu(ctx, idx, mo, q, n_threads)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        """k steps = k mini-batches of the same 1000 proteins, handed to the library as a stream (mpa_map_batches -- what
        miniprot's own batch loop, map.c:273-345, would call): the host seeding of batch i+1 and the formatting of batch i-1
        overlap the GPU DP rounds of batch i, and two batches are in their DP rounds at a time (the rounds of one batch are
        dominated by the tails of a few very long DP calls).  Every step's work, pipeline fill and drain included, happens
        inside the timed region.  --no-stream: one blocking mpa_map_batch() + format per step."""
        if k <= 0:
            return b""
        if a.no_stream:
            out = b""
            for _ in range(k):
                out = step()
            return out
        return mpa.map_batches(ctx, idx, mo, [q] * k, n_threads)[-1]

    paf = run_steps(a.warmup)
    ctx.dp_stats(total=True, reset=True)
    sync()
    t1 = time.time()
    paf = run_steps(a.steps)
    sync()
    elapsed = time.time() - t1
    st = ctx.dp_stats(total=True)                         # accumulated over the DP rounds of the timed steps
    dp_ms = {"prep": st["ms_prep"], "ext": st["ms_ext"], "glob": st["ms_glob"], "backtrack": st["ms_backtrack"]}
    ext_launches, ext_bytes, ext_cells, glob_cells = st["launches_ext"], st["alg_bytes_ext"], st["cells_ext"], st["cells_glob"]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        all_paf = mdist.gather_bytes(paf, rank, world, torch.device("cuda", local_rank))   # the final hit gather (RCCL)
    else:
        all_paf = [paf]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    residues = sum(aligned_residues(p) for p in all_paf)
    value = residues * a.steps / elapsed
    ms_per_step = elapsed / a.steps * 1e3

    # ---- roofline of the dominant kernel (k_ext: >80 % of all DP cells), from HIP events on the executor's stream
    roofline = None
    if ext_launches > 0 and dp_ms["ext"] > 0:
        avg_ms = dp_ms["ext"] / ext_launches
        achieved = ext_bytes / ext_launches / (avg_ms * 1e-3) / 1e9
        gcups = ext_cells / (dp_ms["ext"] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_ext_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "k_ext_narrow + k_ext_wide<NW>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches": ext_launches, "avg_launch_ms": avg_ms,
                    "alg_bytes_per_launch": ext_bytes / ext_launches,
                    "valu_int16": {"gcups": gcups, "ops_per_cell": OPS_PER_CELL_EXT, "achieved_ops_per_s": gcups * 1e9 * OPS_PER_CELL_EXT,
                                   "peak_ops_per_s": VALU_I16_PEAK, "frac": gcups * 1e9 * OPS_PER_CELL_EXT / VALU_I16_PEAK},
                    "note": "score-only DP is VALU/cross-lane bound, not HBM bound (SURVEY.md 8(d)); both fractions reported"}

    # ---- CPU baseline: the reference itself (oracle/_ref/miniprot, built from /root/reference) on a bounded sample
    cpu = None
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "miniprot")
    if world == 1 and not a.no_cpu_baseline and a.cpu_sample > 0 and os.path.exists(ref_bin):   # (rank 0 at N=1 only)
        tmp = "/tmp/mpa_bench_%d" % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        mpi = os.path.join(tmp, "g.mpi")
        idx.dump(mpi)
        ns = min(a.cpu_sample, len(prots0))
        gen_synth.write_fasta_aa(os.path.join(tmp, "p.fa"), prots0[:ns], names0[:ns])
        # the timed sample: the whole protein set, replicated so that the run lasts seconds on a many-core host
        reps = max(1, a.cpu_replicas)
        gen_synth.write_fasta_aa(os.path.join(tmp, "rep.fa"), list(prots0) * reps, ["r%d_%s" % (k, n) for k in range(reps) for n in names0])
        open(os.path.join(tmp, "empty.fa"), "w").close()
        def timed(args, n_runs):
            best, out = 1e30, b""
            for _ in range(n_runs):
                tl = time.time()
                out = subprocess.run([ref_bin, "-t%d" % ncpu, "-u", mpi] + args, capture_output=True).stdout
                best = min(best, time.time() - tl)
            return best, out
        t_load, _ = timed([os.path.join(tmp, "empty.fa")], 3)
        _, out = timed([os.path.join(tmp, "p.fa")], 1)
        t_run, out_rep = timed([os.path.join(tmp, "rep.fa")], 2)
        ours = b"".join(l + b"\n" for l in all_paf[0].split(b"\n")[:-1] if l.split(b"\t", 1)[0].decode() in set(names0[:ns]))
        cpu_res = aligned_residues(out_rep)
        cpu = {"value": cpu_res / max(t_run - t_load, 1e-6), "unit": "aligned residues/s", "cores": ncpu, "kind": "reference",
               "sample": "the %d proteins x %d replicas (%d residues aligned), miniprot -t%d -u on the same .mpi, best of 2; wall %.2fs minus index load %.2fs"
                         % (len(prots0), reps, cpu_res, ncpu, t_run, t_load),
               "paf_identical_on_sample": bool(ours == out), "identity_sample": "first %d proteins" % ns}
        subprocess.run(["rm", "-rf", tmp])

    rec = {"metric": "aligned protein residues/sec (whole node)", "value": value, "unit": "aligned residues/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "int16", "data": "synthetic",
           "config": {"workload": "synthetic %.0f Mbp genome x %d proteins per GPU (~400 aa, 15%% diverged), index resident in HBM, flags -u" % (a.genome_mb, a.n_prot),
                      "proteins_per_gpu": a.n_prot, "aligned_residues_per_step": residues, "host_threads_per_rank": n_threads, "parallelism": "proteins sharded x%d" % world, "batching": "blocking call per step" if a.no_stream else "stream of mini-batches (mpa_map_batches)"},
           "roofline": roofline, "cpu_baseline": cpu,
           "stage_ms_per_step": {k: v / a.steps for k, v in dp_ms.items()},
           "dp_cells_per_step": {"ext": ext_cells // max(a.steps, 1), "glob": glob_cells // max(a.steps, 1)},
           "setup_s": t_setup}
    print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
