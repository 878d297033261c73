#!/usr/bin/env python3
"""bench.py -- aligned protein residues/sec of the MI355X-native miniprot hot path on BASELINE.json's headline workload.

Workload (default, `--config 3` = BASELINE.json configs[2], the one `metric` is quoted on): synthetic 3 Gbp human-like genome
(24 contigs x 125 Mbp, 5 % N runs, 2 % tandem pseudo-paralogs of planted genes) x 20 000 proteins (~400 aa, 15 % diverged),
flags `-I -u` (max_intron from the genome size, options.c:31-37), generator miniprot_amd/synth.py, index
built once and resident in HBM before the timed region.  The 20 000 proteins are 5 distinct mini-batches of 4 000 (the
reference's own mini-batch is 2 M residues, about 5 000 of these proteins: main.c -K).

A "step" is one pass of the whole per-query hot path (seed -> chain -> refine -> spliced DP on the GPU -> CIGAR/statistics ->
PAF text) over THE WHOLE WORKLOAD the metric is quoted on: all 20 000 proteins, as 5 mini-batches of 4 000.  The K timed steps are
K x 5 mini-batches handed to mpa_map_batches() as one stream (what miniprot's own batch loop, map.c:273-345, would call):
pipeline fill and drain are inside the timed region (about 3 % of it at the driver's K = 20; `ms_per_mini_batch` is the old
per-mini-batch figure).  Every output line of every timed step is compared with the reference's output for the same proteins
(`paf_identical`).

    python bench.py                                    # N=1, config 3, 16 steps (16 passes over the 20 000 proteins)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N

N>1 (BASELINE.json configs[3]).  `--scaling weak` (default, what the task's contract asks of a path that shards without a
data-path collective): per-GPU work is fixed, the job is N x K passes.  `--scaling strong` (configs[3] as written: the SAME 20 000
proteins x K passes over 1/2/4/8 GPUs): the job is K passes whatever N is.  Either way the job's mini-batches (batch j of the job =
distinct mini-batch j mod 5) are shared by the N ranks -- one process per GPU, index replicated, no collective on
the data path.  The deal is dynamic: every rank's pipeline claims the next unclaimed batch of the job from a shared counter whenever
it has room for one (mpa_map_batches_claim + miniprot_amd/dist.py; the work stealing of map.c:264-271, between processes); the
output is gathered to rank 0 over RCCL in job order and every line of it is compared with the reference's (`paf_identical`, at any
N).  `value` = aligned residues of all N x K batches / elapsed.  Rank 0 builds genome and index once; the other ranks restore the
.mpi from /dev/shm.
`--config 2` / `--config 5` run the 50 Mbp x 1k and the long-intron (-G 500k) workloads instead; at N=1 they are also
recorded as extra keys of the config-3 line unless --no-extra.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# The DP executor runs the kernel classes of a round on separate HIP streams; the runtime reads its number of hardware queues
# when it initialises, and torch initialises it before libmpamd.so (which sets the same default when loaded) is even imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
# packed-int16 VALU peak: 256 CU x 4 SIMD x 32 lanes/clk x 2 (packed) x 2.4 GHz  (MI355X_MICROARCH.md: SIMD-32, 2400 MHz)
VALU_I16_PEAK = 256 * 4 * 32 * 2 * 2.4e9
VALU_CYCLES_PER_PK_INST = 4.15   # measured: cycles of a SIMD per wave64 packed-int16 (VOP3P) instruction, 8 resident waves (profiles/r06_valu_rate.txt)
VALU_I16_MEASURED = 256 * 4 * 64 * 2 * 2.4e9 / VALU_CYCLES_PER_PK_INST
OPS_PER_CELL_EXT = 32            # int16 ops per DP cell, score-only loop of the reference
OPS_PER_CELL_GLOB = 65           # with traceback

CONFIGS = {
    2: dict(name="config 2: synthetic 50 Mbp genome (1 contig) x 1 000 proteins, flags -u", genome_mb=50, n_ctg=1, n_prot=1000, seed=12, batch=1000,
            gen=dict(), flags=["-u"]),
    3: dict(name="config 3: synthetic 3 Gbp human-like genome (24 contigs x 125 Mbp, 5 % N runs, 2 % tandem pseudo-paralogs) x 20 000 proteins, flags -I -u",
            genome_mb=3000, n_ctg=24, n_prot=20000, seed=3, batch=4000,
            gen=dict(mu=8.0, sigma=1.6, imax=190000, n_frac=0.05, paralog_frac=0.02), flags=["-I", "-u"]),
    5: dict(name="config 5: long-intron stress, 3 Gbp genome x 2 000 proteins with >= 4 exons, introns to 450 kb, flags -G 500k -u",
            genome_mb=3000, n_ctg=24, n_prot=2000, seed=3, batch=500,
            gen=dict(mu=10.5, sigma=1.2, imin=1000, imax=450000, min_exons=4, n_frac=0.05, paralog_frac=0.02), flags=["-G", "500000", "-u"]),
}


def aligned_residues(paf):
    n = 0
    for line in paf.split(b"\n"):
        f = line.split(b"\t", 6)
        if len(f) > 5 and f[5] != b"*":
            n += int(f[3]) - int(f[2])
    return n


def mapopt_for(mpa, idx, flags):
    import ctypes as C
    mo = mpa.default_mapopt()
    it = iter(flags)
    for f in it:
        if f == "-u":
            mo.flag |= 0x4
        elif f == "-I":
            mpa.lib().mpa_mapopt_set_max_intron(C.byref(mo), idx.genome_len())
        elif f == "-G":
            mo.max_intron = mo.bw = int(next(it))
    return mo


class Workload:
    """Genome index + protein mini-batches of one config.  Rank 0 generates and indexes; with several ranks the index and the
    proteins travel through a directory in /dev/shm (mpa_idx_dump / mpa_idx_restore), so the host work is done once."""

    def __init__(self, mpa, cfg, rank, world, barrier, share_dir, n_threads, scale=1.0, ctx=None):
        from miniprot_amd import synth
        self.cfg, self.mpa = cfg, mpa
        t0 = time.time()
        mpi, faa = os.path.join(share_dir, "g.mpi"), os.path.join(share_dir, "p.faa")
        self.mpi, self.faa = mpi, faa
        cached = os.path.exists(mpi) and os.path.exists(faa)        # (--workload-cache: a previous run on this box left them)
        if rank == 0 and not cached:
            n_prot = max(10, int(cfg["n_prot"] * scale))
            contigs, prots, names = synth.generate(int(cfg["genome_mb"] * 1e6 * scale), cfg["n_ctg"], n_prot, cfg["seed"], **cfg["gen"])
            self.t_generate = time.time() - t0
            t1 = time.time()
            self.idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
            del contigs
            # the k-mer table: scan + one radix sort on the GPU (mpa_idx_build_kmers_device), on the host cores if that declines
            self.index_built_on = self.idx.build_kmers(max(2, n_threads), ctx)
            self.t_index = time.time() - t1
            synth.write_fasta_aa(faa, prots, names)
            self.idx.dump(mpi)                                # for the other ranks and for the reference binary
        barrier()
        if rank != 0 or cached:
            self.idx = mpa.Index.restore(mpi)
            names, prots = [], []
            for line in open(faa, "rb"):
                (names if line.startswith(b">") else prots).append(line[1:].strip().decode() if line.startswith(b">") else line.strip())
            self.t_generate = self.t_index = 0.0
            self.index_built_on = "cache" if cached else "rank 0"
        self.prots, self.names = prots, names
        b = cfg["batch"]
        self.batches = [mpa.Queries(prots[i:i + b], names[i:i + b]) for i in range(0, len(prots), b)]
        self.batch_of = {}
        for k, q in enumerate(self.batches):
            for n in q.names:
                self.batch_of[n.encode()] = k
        self.mo = mapopt_for(mpa, self.idx, cfg["flags"])
        self.t_setup = time.time() - t0

    def reference(self, ncpu, ncpu_eff=None):
        """oracle/_ref/miniprot (the reference itself, compiled from /root/reference) on the whole protein set, same .mpi, all
        host cores; wall time minus an index-load-only run.  Where the box's CPU quota is below the cores it shows, the reference
        is run with -t<quota> as well and the faster of the two runs is the baseline.  Returns (record, per-batch reference text)."""
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "miniprot")
        if not os.path.exists(ref_bin):
            return None, None
        empty = os.path.join(os.path.dirname(self.faa), "empty.faa")
        open(empty, "w").close()
        def timed(q, nt):
            t = time.time()
            out = subprocess.run([ref_bin, "-t%d" % nt] + self.cfg["flags"] + [self.mpi, q], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            return time.time() - t, out
        t_load = min(timed(empty, ncpu)[0] for _ in range(2))
        t_run, out = timed(self.faa, ncpu)
        tried = {ncpu: t_run}
        if ncpu_eff and ncpu_eff < ncpu:
            t2, out2 = timed(self.faa, ncpu_eff)
            tried[ncpu_eff] = t2
            if t2 < t_run:
                t_run, out, ncpu = t2, out2, ncpu_eff
        per = [[] for _ in self.batches]
        for line in out.split(b"\n"):
            if line:
                per[self.batch_of[line.split(b"\t", 1)[0]]].append(line + b"\n")
        per = [b"".join(x) for x in per]
        res = aligned_residues(out)
        # the same binary on ONE thread, on a bounded sample of the same protein set (the per-core rate next to the
        # all-cores one): the first proteins of the set, about 10-20 s of CPU work
        t1 = None
        try:
            n1 = max(1, min(len(self.prots), int(self.cfg.get("t1_sample", 800))))
            sub = os.path.join(os.path.dirname(self.faa), "t1_sample.faa")
            with open(self.faa) as f, open(sub, "w") as g:
                k = 0
                for line in f:
                    if line.startswith(">"):
                        k += 1
                        if k > n1:
                            break
                    g.write(line)
            t_load1 = timed(empty, 1)[0]
            t_one, out1 = timed(sub, 1)
            res1 = aligned_residues(out1)
            t1 = {"value": res1 / max(t_one - t_load1, 1e-6), "unit": "aligned residues/s", "cores": 1,
                  "sample": "the first %d proteins of the set, -t1, wall %.2f s minus index load %.2f s; %d residues aligned" % (n1, t_one, t_load1, res1)}
        except Exception as e:                                       # (the one-thread figure is a courtesy: never fail the bench for it)
            t1 = {"error": repr(e)[:200]}
        rec = {"value": res / max(t_run - t_load, 1e-6), "unit": "aligned residues/s", "cores": ncpu, "kind": "reference", "cpu_model": cpu_model(), "one_thread": t1,
               "sample": "the whole workload: %d proteins, oracle/_ref/miniprot -t%d %s on the same .mpi; wall %.2f s minus index load %.2f s; %d residues aligned"
                         % (len(self.prots), ncpu, " ".join(self.cfg["flags"]), t_run, t_load, res),
               "wall_s": t_run, "index_load_s": t_load, "md5": hashlib.md5(out).hexdigest(), "wall_s_by_threads": {str(k): v for k, v in tried.items()}}
        return rec, per


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def host_cpu_seconds():
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def cgroup_throttled():
    """(periods throttled, seconds throttled) of this process's cgroup so far: a CPU quota below the visible cores freezes ALL
    threads of the process whenever a 100 ms period's budget is spent"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e6
    except Exception:
        return 0, 0.0


def run_stream(mpa, mdist, wl, ctx, n_batches, n_threads, rank, world, device):
    """One job of n_batches mini-batches (batch k of the job = distinct batch k mod n_distinct), shared by all ranks: every rank's
    mpa_map_batches_claim() stream pulls the next unclaimed batch whenever its pipeline has room (miniprot_amd/dist.py), the
    text of every batch goes to rank 0.  Returns the texts in job order on rank 0."""
    nb = len(wl.batches)
    job = [wl.batches[k % nb] for k in range(n_batches)]
    def map_claimed(claim):
        return mpa.map_batches(ctx, wl.idx, wl.mo, job, n_threads, claim=claim)
    texts, _ = mdist.run_sharded(n_batches, rank, world, device, False, None, None, None, map_claimed=map_claimed)
    return texts


def measure(mpa, mdist, wl, ctx, steps, warmup, n_threads, rank, world, device, sync, weak):
    # a step = one pass over the workload = len(wl.batches) mini-batches; weak scaling: `steps` and `warmup` are per GPU (the job
    # the ranks share is world x steps passes), strong scaling: the job is `steps` passes whatever the world size
    nb = len(wl.batches)
    mult = world if weak else 1
    if warmup > 0:
        run_stream(mpa, mdist, wl, ctx, warmup * nb * mult, n_threads, rank, world, device)
    ctx.dp_stats(total=True, reset=True)
    mpa.Context.stage_clocks(reset=True)
    grow0 = mpa.Context.pool_growths()
    sync()
    cpu0 = host_cpu_seconds()
    thr0 = cgroup_throttled()
    t1 = time.time()
    texts = run_stream(mpa, mdist, wl, ctx, steps * nb * mult, n_threads, rank, world, device)
    sync()
    elapsed = time.time() - t1
    st = ctx.dp_stats(total=True)
    st["stage_clocks"] = mpa.Context.stage_clocks()
    st["pool_growths_timed"] = mpa.Context.pool_growths() - grow0          # device pools re-allocated inside the timed region (each one a device-wide hipFree)
    st["pool_growths_total"] = mpa.Context.pool_growths()
    st["hbm_bytes"] = mpa.Context.device_bytes()
    st["host_cpu_s"] = host_cpu_seconds() - cpu0              # user + system CPU time of THIS rank's process (all its threads) inside the timed region
    thr1 = cgroup_throttled()
    st["throttled_periods"], st["throttled_s"] = thr1[0] - thr0[0], thr1[1] - thr0[1]
    return elapsed, texts, st


def roofline_record(st, elapsed, world):
    """Roofline of the dominant kernel.  With one launch per DP round (k_dp_round, the default) that kernel does every
    extension call and the traceback calls of the round's first chunk; its duration is measured with HIP events on the stream
    it is launched on.  (With MPA_ONE_LAUNCH=0 the figures are those of the per-class extension kernels, whose durations
    overlap.)"""
    pool = None
    if st["launches_round"] > 0 and st["ms_round"] > 0 and st.get("ms_round_union", 0) > 0:
        # DP worker pool (MPA_DP_POOL=1): the units of all rounds in flight are served by ONE population of k_dp_worker workgroups; a worker
        # launch (one per round, from the lane's worker stream) runs units of any lane, so "ops of a launch" exist only in sum.
        # The kernel's duration = the time during which the pool has a launch running (union of the launches' HIP-event intervals):
        # what ONE persistent kernel's duration would be.  Sum and mean of the launches themselves are kept next to it (`per_launch`:
        # the figures rocprofv3's kernel trace shows for k_dp_worker; they overlap up to MPA_DP_LANES deep).
        kernel = "k_dp_worker (DP worker pool: every extension class + first traceback chunk of all rounds in flight; one launch of workers per round)"
        n, ms = st["launches_round"], st["ms_round_union"]
        ops = OPS_PER_CELL_EXT * st["cells_ext_round"] + OPS_PER_CELL_GLOB * st["cells_glob_round"]
        cells = st["cells_ext_round"] + st["cells_glob_round"]
        pool = {"launches": n, "sum_launch_ms": st["ms_round"], "avg_launch_ms": st["ms_round"] / n, "union_ms": ms,
                "frac_by_sum_of_launches": ops / (st["ms_round"] * 1e-3) / VALU_I16_PEAK,
                "note": "achieved/frac above use union_ms (time with >= 1 worker launch running on the device); frac_by_sum_of_launches divides by the "
                        "SUM of the overlapping launches' durations instead"}
        note = "HIP events on the lanes' worker streams against one base event; ops = all units of the timed rounds"
    elif st["launches_round"] > 0 and st["ms_round"] > 0:
        kernel = "k_dp_round (one launch per DP round: all extension classes + first traceback chunk)"
        n, ms = st["launches_round"], st["ms_round"]
        ops = OPS_PER_CELL_EXT * st["cells_ext_round"] + OPS_PER_CELL_GLOB * st["cells_glob_round"]
        cells = st["cells_ext_round"] + st["cells_glob_round"]
        note = "launch durations from HIP events on the launch stream; the rounds of the batches in flight (MPA_DP_LANES) overlap on the GPU"
    elif st["launches_ext"] > 0 and st["ms_ext"] > 0:
        kernel = "k_ext_narrow + k_ext_wide<NW> + k_ext_wide_split (score-only extension sweep, > 80 % of all DP cells)"
        n, ms = st["launches_ext"], st["ms_ext"]
        ops, cells = OPS_PER_CELL_EXT * st["cells_ext"], st["cells_ext"]
        note = "per-launch durations are HIP-event times of kernels that run CONCURRENTLY on several streams (sum over launches > wall time)"
    else:
        return None
    ach = ops / (ms * 1e-3)                                   # int16 ops/s while the kernel is running
    # algorithmic HBM bytes of the launch: extension calls = packed window + protein + result; traceback calls add
    # 2 bytes per cell written, the words read along the path and the CIGAR -- the share of the traceback calls that ride in the
    # round kernel is taken by cells
    glob_share = st["cells_glob_round"] / st["cells_glob"] if st.get("cells_glob") else 0.0
    alg_bytes = st["alg_bytes_ext"] + st.get("alg_bytes_glob", 0) * glob_share
    hbm = alg_bytes / (ms * 1e-3) / 1e9
    wall_ops = (OPS_PER_CELL_EXT * st["cells_ext"] + OPS_PER_CELL_GLOB * st["cells_glob"]) / elapsed
    traffic, src = None, None
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs, tools/pmc_round.sh): a
    # profile constant measured on this workload at this mini-batch size, NOT collected in this run
    # -- and only quoted when it was measured on THIS source of the DP kernels (the profile carries the md5 of dp_kernels.hip +
    # dp_device.h it was collected with): counters of another kernel next to this run's timings would be a stale claim
    cur = hashlib.md5()
    for f in ("dp_kernels.hip", "dp_device.h"):
        try:
            cur.update(open(os.path.join(ROOT, "miniprot_amd", "csrc", f), "rb").read())
        except OSError:
            pass
    cur = cur.hexdigest()
    kname = "k_dp_worker" if pool else "k_dp_round"
    valu_share = None
    for name, what in (("r06_pmc_summary.json", "config 3 at full size, this mini-batch size, two DP rounds per batch"),
                       ("r05_pmc_summary.json", "config 3 at full size, this mini-batch size, two DP rounds per batch")):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                d = json.load(open(pmc))
                if d.get("dp_kernel_source_md5") != cur:
                    src = "profiles/%s was collected on another source of the DP kernels (md5 %s, running %s): not quoted" % (name, str(d.get("dp_kernel_source_md5"))[:8], cur[:8])
                    continue
                traffic = d["kernels"][kname]["hbm_bytes_per_launch"]
                k = d["kernels"][kname]
                if k.get("valu_insts_per_launch") and k.get("avg_us"):
                    # share of the chip's VALU issue cycles the launch used, at the MEASURED 4.15 cycles per packed instruction
                    # (profiles/r06_valu_rate.txt), over the launch's whole duration (the PMC passes run the launch alone)
                    valu_share = k["valu_insts_per_launch"] * VALU_CYCLES_PER_PK_INST / (256 * 4 * k["avg_us"] * 1e-6 * 2.4e9)
                src = "profiles/%s: separate rocprofv3 --pmc passes (tools/pmc_round.sh) over %s; kernel source md5 %s = the running library's" % (name, what, cur[:8])
                break
            except Exception:
                pass
    return {"kernel": kernel, "bound": "valu-int16", "achieved": ach / 1e12, "peak": VALU_I16_PEAK / 1e12, "unit": "Tops/s (packed int16)", "frac": ach / VALU_I16_PEAK,
            "traffic": traffic, "traffic_source": src,
            # the same achieved rate against what the chip was MEASURED to issue: a packed int16 instruction takes 4.15 cycles of its
            # SIMD, not the 2 the guide's wave64 rate (and `peak`) assume (tools/ubench/valu_rate.hip, profiles/r06_valu_rate.txt)
            "measured_issue_peak": VALU_I16_MEASURED / 1e12, "frac_of_measured_issue_peak": ach / VALU_I16_MEASURED,
            "wall_frac": wall_ops / VALU_I16_PEAK / max(world, 1), "valu_issue_share_over_launch": valu_share,
            "launches": n, "avg_launch_ms": (st["ms_round"] if pool else ms) / n, "kernel_active_ms": ms, "cells_per_launch": cells / n, "ops_per_cell": {"score_only": OPS_PER_CELL_EXT, "traceback": OPS_PER_CELL_GLOB},
            "gcups_per_launch": cells / (ms * 1e-3) / 1e9,          # (worker pool: GCUPS while the pool is running)
            "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "alg_bytes_per_launch": alg_bytes / n,
                    "note": "algorithmic bytes of the launch's calls (SURVEY.md 8(d): extension = packed window + protein + result; traceback adds 2 B per cell, the path's words and the CIGAR) / launch time: not the binding roof"},
            "wall": {"achieved": wall_ops / 1e12, "frac": wall_ops / VALU_I16_PEAK / max(world, 1), "unit": "Tops/s",
                     "note": "all DP ops of the timed steps (32/cell score-only + 65/cell traceback) / elapsed wall time / GPUs: includes every host stage and idle gap"},
            "per_launch": pool, "note": note}


def cli_end_to_end(wl, n_threads, cpu):
    """The drop-in as a user runs it, cold: the reference's own main.c relinked against libmpamd.so (examples/_relink/miniprot)
    on the same .mpi and protein file as the reference binary -- whole wall clock including index restore, upload to HBM, pool
    growth and pipeline fill (main.c:189-217, index.c:204-237) -- next to oracle/_ref/miniprot's whole wall clock, bytes compared."""
    exe = os.path.join(ROOT, "examples", "_relink", "miniprot")
    if not os.path.exists(exe) or cpu is None:
        return None
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "miniprot_amd") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.setdefault("MPA_DEVICES", "0")          # one GPU, like the line's own n_gpus (the file mapper's own default is one device as well)
    cmd = [exe, "-t%d" % n_threads] + wl.cfg["flags"] + [wl.mpi, wl.faa]
    t = time.time()
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t
    if pr.returncode != 0:
        return {"error": "exit %d: %s" % (pr.returncode, pr.stderr.decode(errors="replace")[-300:])}
    md5 = hashlib.md5(pr.stdout).hexdigest()
    return {"command": " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd), "wall_s": wall, "reference_wall_s": cpu["wall_s"],
            "reference_threads": cpu["cores"], "devices": env["MPA_DEVICES"], "speedup_end_to_end": cpu["wall_s"] / max(wall, 1e-9), "md5": md5, "identical_to_reference": md5 == cpu["md5"],
            "note": "one cold run each, index load included on both sides (reference: %.1f s of its wall)" % cpu["index_load_s"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed steps; one step = one pass over the config's whole protein set (config 3: 5 mini-batches of 4 000)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps (passes) before the timed ones")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="N > 1: weak = every GPU brings its own --steps passes (job = N x steps); "
                    "strong = BASELINE configs[3] as written, the same --steps passes shared by all GPUs")
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--scale", type=float, default=1.0, help="(debug) shrink genome and protein set by this factor; the line is then NOT the named config")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the non-DP stages (0 = cores / ranks, at most 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="proteins per mini-batch (0 = the config's own)")
    ap.add_argument("--workload-cache", default="", help="(debug) keep the generated genome index and proteins in this directory and reuse them if present")
    ap.add_argument("--no-extra", action="store_true", help="N=1, config 3: skip the config-2 and config-5 records")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end run of the relinked miniprot command line (cli_end_to_end)")
    ap.add_argument("--debug-shared-gpu", action="store_true", help="(debug) all ranks on cuda:0 with the gloo backend: exercises the N>1 code path on a "
                    "one-GPU box; the line is NOT a scaling measurement")
    a = ap.parse_args()

    # --gpus N is a promise about the line's n_gpus.  Started plainly (no RANK in the environment) with N > 1 the script launches
    # its own N ranks -- one process per GPU, torch.distributed.run on this node -- and started under a launcher it insists that the
    # launcher's world is the N that was asked for; either way fewer than N visible devices is an error, not a smaller run.
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "RANK" not in os.environ and a.gpus > 1:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus and not a.debug_shared_gpu:
            raise SystemExit("bench.py: --gpus %d asked for, %d GPU(s) visible" % (a.gpus, have))
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s rank(s)" % (a.gpus, os.environ.get("WORLD_SIZE", "1")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if "RANK" in os.environ:                                # launched by torch.distributed.run (any world size)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.debug_shared_gpu:
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DP kernels have no CPU fallback")
    if not a.debug_shared_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d rank(s), %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cpu") if a.debug_shared_gpu else torch.device("cuda", local_rank)

    import miniprot_amd as mpa
    from miniprot_amd import dist as mdist

    ncpu = os.cpu_count() or 1
    # what the box really gives this process: a cgroup CPU quota (cpu.max = "quota period") can be far below the cores it shows --
    # 16 of 256 on the boxes this was developed on -- and threads beyond it only get the whole process throttled
    ncpu_eff = ncpu
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncpu_eff = max(1, min(ncpu, int(round(int(q) / int(per)))))
    except Exception:
        pass
    # host threads of the planning stage (the other pipeline stages take a quarter each): the host stages are memory-bound and
    # more than 64 threads lose (measured at config 3 on the 256-thread host: 32 -> 0.98 s, 64 -> 1.00 s, 96 -> 1.10 s, 128 -> 1.22 s)
    # (a rank's pipeline runs about 2.75 x n_threads host threads: two planners with half each, two seeders, four DP lanes and
    # the output stage with a quarter each -- so N ranks share the host's hardware threads without oversubscribing them)
    # (round 6, 16-core quota, same-lease repeats: 12 threads 22.4-23.7 M residues/s, 24: 24.0-25.2, 32: 24.2-24.4, 40: 25.0-26.2, 56: 24.8-25.6 --
    # the host stages are short bursts on the batches' critical path, 7 cores busy on average: 2.5 x the quota)
    n_threads = a.threads if a.threads > 0 else max(4, min(64, int(ncpu / max(world, 1) / 2.75), int(2.5 * ncpu_eff / max(world, 1))))

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        barrier()
        torch.cuda.synchronize()

    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    share_root = os.path.join(base, "mpa_bench_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))

    def run_config(cid, steps, warmup, with_cpu, batch=0, keep_share=False):
        cfg = dict(CONFIGS[cid])
        if batch > 0:
            cfg["batch"] = batch
        elif a.batch > 0 and cid == a.config:
            cfg["batch"] = a.batch
        share = os.path.join(a.workload_cache or share_root, "c%d%s" % (cid, "" if a.scale == 1.0 else "_x%g" % a.scale))
        if rank == 0:
            os.makedirs(share, exist_ok=True)
        barrier()
        ctx = mpa.Context(local_rank)
        wl = Workload(mpa, cfg, rank, world, barrier, share, n_threads, a.scale, ctx)
        wl.idx.to_device(ctx)
        weak = a.scaling == "weak"
        elapsed, texts, st = measure(mpa, mdist, wl, ctx, steps, warmup, n_threads, rank, world, device, sync, weak)
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            keys = ("cells_ext", "cells_glob", "alg_bytes_ext", "alg_bytes_glob", "launches_ext", "ms_ext", "cells_ext_round", "cells_glob_round", "launches_round", "ms_round", "ms_round_union", "host_cpu_s", "hbm_bytes")
            tot = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64, device=device)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            for k, v in zip(keys, tot.tolist()):
                st[k] = v
        rec = None
        if rank == 0:
            nb = len(wl.batches)
            res_of = {}
            residues = 0
            for k, t in enumerate(texts):
                if k % nb not in res_of or k < nb:
                    res_of[k % nb] = aligned_residues(t)
                residues += res_of[k % nb]
            one_pass = b"".join(texts[k] for k in range(min(nb, len(texts))))
            cpu, identical, ident = None, None, None
            if with_cpu and not a.no_cpu_baseline:
                # the reference itself on this box's host cores, outside the timed region: every output line of every timed step of
                # every rank is compared with it (at any N); its timing is the CPU baseline, reported at N = 1 only
                cpu, per = wl.reference(ncpu, ncpu_eff)
                if cpu is not None:
                    bad = [k for k, t in enumerate(texts) if t != per[k % nb]]
                    identical = not bad
                    ident = {"paf_identical": identical, "lines_compared": sum(t.count(b"\n") for t in texts), "first_differing_step": bad[0] if bad else None,
                             "reference_md5": cpu["md5"]}
                    cpu.update(ident)
                    if world > 1:
                        cpu = None
            sc = st["stage_clocks"]
            per = lambda k: (sc[k][0] / sc[k][1]) if sc[k][1] else None
            cli = None                                                # (filled in below, once this process has given its HBM back)
            rec = {"workload": cfg["name"] + ("" if a.scale == 1.0 else " [SCALED x%g: not the named config]" % a.scale),
                   "value": residues / elapsed, "elapsed_s": elapsed, "steps": steps, "mini_batches_all_gpus": len(texts),
                   # per GPU under weak scaling (every GPU runs `steps` passes), per job under strong scaling (the job is `steps` passes)
                   "ms_per_step": elapsed / max(steps, 1) * 1e3, "ms_per_step_is": "per GPU" if weak else "per job",
                   "ms_per_mini_batch": elapsed / max(len(texts) / max(world, 1), 1) * 1e3,
                   "proteins_per_step": len(wl.prots), "proteins_per_mini_batch": cfg["batch"], "mini_batches_per_step": nb, "aligned_residues": residues,
                   # how many k_dp_round launches are in flight on a GPU on average: sum of the launch durations / elapsed
                   "rounds_in_flight_avg": st["ms_round"] * 1e-3 / elapsed / max(world, 1),
                   # wall clock of the pipeline stages per mini-batch (rank 0; stage threads run side by side: these are latencies)
                   "seeding_wall_ms_per_batch": per("seeding"), "planning_wall_ms_per_batch": per("planning"), "dp_wall_ms_per_batch": per("dp"),
                   "output_wall_ms_per_batch": per("output"), "sketch_wall_ms_per_batch": per("sketch"),
                   # HBM held through the library (resident index + all pools of all contexts), per GPU; pool re-allocations inside the timed region
                   "hbm_resident_gb": st["hbm_bytes"] / max(world, 1) / 1e9, "pool_growth_events_timed": st["pool_growths_timed"],
                   "pool_growth_events_total": st["pool_growths_total"], "cli_end_to_end": cli,
                   "input_residues_per_s": sum(len(p) for k in range(len(texts)) for p in wl.batches[k % nb].seqs) / elapsed,
                   "md5_first_pass": hashlib.md5(one_pass).hexdigest() if len(texts) >= nb else None, "paf_identical": identical, "identity": ident,
                   "roofline": roofline_record(st, elapsed, world), "cpu_baseline": cpu,
                   "dp_cells": {"ext": st["cells_ext"], "glob": st["cells_glob"]},
                   "gcups_wall": (st["cells_ext"] + st["cells_glob"]) / elapsed / 1e9,
                   "setup_s": {"generate": wl.t_generate, "index_build": wl.t_index, "index_built_on": wl.index_built_on, "total": wl.t_setup},
                   "host_threads_per_rank": n_threads, "host_cpus_visible": ncpu, "host_cpu_quota": ncpu_eff,
                   # CPU seconds (user + system, getrusage) all ranks' processes spent inside the timed region, per step and as a share of
                   # the host: what one GPU's pipeline costs the host it shares with the other ranks
                   "host_cpu_s_per_step": st["host_cpu_s"] / max(len(texts), 1) * nb, "host_cpu_s_per_mini_batch": st["host_cpu_s"] / max(len(texts), 1),
                   "host_cores_busy": st["host_cpu_s"] / max(elapsed, 1e-9),
                   # what a rank's pipeline keeps busy on the host (user + system CPU seconds of its process / wall): N ranks need N x this
                   "host_cores_busy_per_gpu": st["host_cpu_s"] / max(elapsed, 1e-9) / max(world, 1),
                   # (rank 0's cgroup: 100 ms scheduler periods in which the quota ran out, and thread-seconds spent frozen because of it)
                   "host_throttled": {"periods": st.get("throttled_periods"), "thread_seconds": st.get("throttled_s")}}
        ctx.close()
        wl.idx.close()
        if rec is not None and rec["cpu_baseline"] is not None and world == 1 and a.scale == 1.0 and not a.no_cli and cid == a.config:
            try:
                rec["cli_end_to_end"] = cli_end_to_end(wl, ncpu_eff, rec["cpu_baseline"])
            except Exception as e:                               # the extra record must not cost the headline line
                rec["cli_end_to_end"] = {"error": repr(e)[:300]}
        barrier()
        if rank == 0 and not a.workload_cache and not keep_share:
            shutil.rmtree(share, ignore_errors=True)
        return rec

    want_extra = world == 1 and a.config == 3 and not a.no_extra and a.scale == 1.0
    main_rec = run_config(a.config, a.steps, a.warmup, with_cpu=True, keep_share=want_extra and a.batch == 0)
    extra = {}
    if want_extra and a.batch == 0:
        # the same workload (index and proteins of the headline run, restored from its share directory) with 8 000 proteins per
        # mini-batch: a DP round lasts as long as its longest window whatever the batch holds, so the per-launch roofline fraction
        # grows with the mini-batch; the headline keeps 4 000 because the stream is fastest there
        try:
            r = run_config(3, 4, 1, with_cpu=False, batch=8000)
            extra["config3_batch8000"] = {x: r[x] for x in ("workload", "value", "elapsed_s", "steps", "ms_per_step", "ms_per_mini_batch", "proteins_per_mini_batch", "gcups_wall",
                                                             "md5_first_pass", "hbm_resident_gb", "rounds_in_flight_avg")}
            extra["config3_batch8000"]["md5_equals_headline"] = r["md5_first_pass"] == main_rec["md5_first_pass"] if rank == 0 else None
            extra["config3_batch8000"]["roofline_frac"] = r["roofline"]["frac"] if r["roofline"] else None
            extra["config3_batch8000"]["avg_launch_ms"] = r["roofline"]["avg_launch_ms"] if r["roofline"] else None
        except Exception as e:                               # an extra must not cost the headline line
            extra["config3_batch8000"] = {"error": repr(e)[:300]}
    if want_extra:
        # (config 2: a pass is ONE mini-batch of 1 000 proteins; config 5: four of 500)
        for cid, (k, w) in ((2, (40, 4)), (5, (3, 1))):
            try:
                r = run_config(cid, k, w, with_cpu=True)
                extra["config%d" % cid] = {x: r[x] for x in ("workload", "value", "elapsed_s", "steps", "ms_per_step", "ms_per_mini_batch", "proteins_per_step", "paf_identical",
                                                              "gcups_wall", "md5_first_pass", "hbm_resident_gb", "rounds_in_flight_avg")}
                extra["config%d" % cid]["cpu_baseline"] = r["cpu_baseline"]
                extra["config%d" % cid]["roofline_frac"] = r["roofline"]["frac"] if r["roofline"] else None
            except Exception as e:                               # an extra must not cost the headline line
                extra["config%d" % cid] = {"error": repr(e)[:300]}
    if rank == 0:
        shutil.rmtree(share_root, ignore_errors=True)
        r = main_rec
        out = {"metric": "aligned protein residues/sec (whole node), 3 Gbp genome x 20k proteins" if a.config == 3 and a.scale == 1.0 else "aligned protein residues/sec (whole node)",
               "value": r["value"], "unit": "aligned residues/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms_per_step"],
               "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "int16", "data": "synthetic",
               "config": {"workload": r["workload"], "proteins_per_step": r["proteins_per_step"], "proteins_per_mini_batch": r["proteins_per_mini_batch"],
                          "mini_batches_per_step": r["mini_batches_per_step"],
                          "parallelism": "one job of %d step(s) x %d mini-batches (%s scaling); every rank's pipeline claims the next unclaimed mini-batch from a shared counter "
                                         "(dynamic deal), index replicated on %d GPU(s), no collective on the data path, text gathered to rank 0 in job order"
                                         % (a.steps * (world if a.scaling == "weak" else 1), r["mini_batches_per_step"], a.scaling, world),
                          "batching": "stream of mini-batches (mpa_map_batches)", "host_threads_per_rank": r["host_threads_per_rank"]},
               "ms_per_step_is": r["ms_per_step_is"], "ms_per_mini_batch": r["ms_per_mini_batch"],
               "elapsed_s": r["elapsed_s"], "aligned_residues": r["aligned_residues"], "input_residues_per_s": r["input_residues_per_s"],
               "paf_identical": r["paf_identical"], "identity": r["identity"], "md5_first_pass": r["md5_first_pass"], "mini_batches_all_gpus": r["mini_batches_all_gpus"],
               "rounds_in_flight_avg": r["rounds_in_flight_avg"], "seeding_wall_ms_per_batch": r["seeding_wall_ms_per_batch"],
               "planning_wall_ms_per_batch": r["planning_wall_ms_per_batch"], "dp_wall_ms_per_batch": r["dp_wall_ms_per_batch"], "output_wall_ms_per_batch": r["output_wall_ms_per_batch"],
               "sketch_wall_ms_per_batch": r["sketch_wall_ms_per_batch"],
               "hbm_resident_gb": r["hbm_resident_gb"], "pool_growth_events_timed": r["pool_growth_events_timed"], "pool_growth_events_total": r["pool_growth_events_total"],
               "cli_end_to_end": r["cli_end_to_end"],
               "host_cpu_s_per_step": r["host_cpu_s_per_step"], "host_cpu_s_per_mini_batch": r["host_cpu_s_per_mini_batch"], "host_cores_busy": r["host_cores_busy"],
               "host_cores_busy_per_gpu": r["host_cores_busy_per_gpu"], "host_throttled": r["host_throttled"],
               "roofline": r["roofline"], "cpu_baseline": r["cpu_baseline"], "dp_cells": r["dp_cells"], "gcups_wall": r["gcups_wall"], "setup_s": r["setup_s"]}
        out.update(extra)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
