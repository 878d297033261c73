"""Several GPUs inside ONE process, through the C boundary: mpa_map_batches_multi() runs one pipeline per device context, the
pipelines claim the job's mini-batches from a shared counter (the kt_for of mp_map_file, map.c:264-271,291, across devices), and
the text comes back in input order with the hit ids of a single pipeline's run.  The box has one GPU: both pipelines sit on
device 0 (two root contexts, two DP worker pools, one resident index) -- the code path a node with eight takes.  The relinked
reference command line takes the same path with MPA_DEVICES=0,0."""
import os
import subprocess
import pytest
import refbind

pytestmark = pytest.mark.gpu


def _job():
    import miniprot_amd as mpa
    import golden
    case = golden.SYNTH_CASES[1]
    contigs, prots, names = golden.synth_inputs(case)
    idx = mpa.Index.from_nt4(contigs, ["chr%d" % (i + 1) for i in range(len(contigs))])
    mpa._check(mpa.lib().mpa_idx_build_kmers(idx.h, 2))
    n = len(prots)
    cuts = [0, n // 7, n // 7, n // 3, n // 2, n - 3, n]               # six uneven mini-batches, one of them empty
    batches = [mpa.Queries(prots[a:b], names[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    return mpa, golden, case, idx, batches, prots, names


def test_two_pipelines_of_one_process_equal_one_pipeline():
    mpa, golden, case, idx, batches, prots, names = _job()
    c0, c1 = mpa.Context(0), mpa.Context(0)
    try:
        for tag, flag in (("paf", 0), ("gff", 0x8), ("gtf", 0x20)):
            mo = golden.mapopt_for(case)
            mo.flag |= flag
            idx.to_device(c0)
            single = b"".join(mpa.map_batches(c0, idx, mo, batches, 2))
            multi = b"".join(mpa.map_batches_multi([c0, c1], idx, mo, batches, 4))
            assert multi == single, tag
            if tag == "paf":                                      # ... which are the reference's bytes
                assert golden.file_header(case) + multi == open(golden.path(case["name"] + ".ref.paf"), "rb").read()
    finally:
        c1.close()
        c0.close()
        idx.close()


def test_relinked_reference_cli_maps_on_a_device_list(tmp_path):
    """The reference's own main.c, relinked (examples/_relink/miniprot), with MPA_DEVICES=0,0: mp_map_file() runs two pipelines and
    prints the bytes of the one-pipeline run."""
    exe = os.path.join(refbind.ROOT, "examples", "_relink", "miniprot")
    if not os.path.exists(exe):
        pytest.skip("examples/_relink/miniprot is built where /root/reference exists")
    mpa, golden, case, idx, batches, prots, names = _job()
    mpi = str(tmp_path / "g.mpi")
    faa = str(tmp_path / "p.faa")
    idx.dump(mpi)
    idx.close()
    with open(faa, "w") as f:
        for nm, p in zip(names, prots):
            f.write(">%s\n%s\n" % (nm, p if isinstance(p, str) else p.decode()))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(refbind.ROOT, "miniprot_amd") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    outs = {}
    for devs in ("0", "0,0"):
        e = dict(env, MPA_DEVICES=devs)
        # -K: mini-batch size in residues -- small, so that the job has several mini-batches for the two pipelines to share
        r = subprocess.run([exe, "-t4", "-K", "20k", "--gff", mpi, faa], env=e, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        outs[devs] = r.stdout
    assert outs["0,0"] == outs["0"]
    assert outs["0"].count(b"\n") > 10
    ref = os.path.join(refbind.ROOT, "oracle", "_ref", "miniprot")
    if os.path.exists(ref):                                   # ... and they are the reference's own bytes
        r = subprocess.run([ref, "-t4", "-K", "20k", "--gff", mpi, faa], capture_output=True, timeout=600)
        assert r.returncode == 0
        assert outs["0,0"] == r.stdout


def test_eight_pipelines_on_one_device(tmp_path):
    """What a node with eight GPUs runs, on the one GPU the box has: MPA_DEVICES=0,0,0,0,0,0,0,0 through the relinked reference
    command line -- eight pipelines (eight root contexts, ~100 stage threads) claiming from one counter: no deadlock, the bytes of
    the one-pipeline run and of the reference, and the host CPU the eight pipelines take (recorded, not asserted)."""
    import time
    import resource
    exe = os.path.join(refbind.ROOT, "examples", "_relink", "miniprot")
    if not os.path.exists(exe):
        pytest.skip("examples/_relink/miniprot is built where /root/reference exists")
    mpa, golden, case, idx, batches, prots, names = _job()
    mpi = str(tmp_path / "g.mpi")
    faa = str(tmp_path / "p.faa")
    idx.dump(mpi)
    idx.close()
    with open(faa, "w") as f:
        for rep in range(4):                                      # enough mini-batches for eight pipelines to share
            for nm, p in zip(names, prots):
                f.write(">%s_%d\n%s\n" % (nm, rep, p if isinstance(p, str) else p.decode()))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(refbind.ROOT, "miniprot_amd") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    outs, cpu = {}, {}
    for devs in ("0", "0,0,0,0,0,0,0,0"):
        e = dict(env, MPA_DEVICES=devs)
        r0, t0 = resource.getrusage(resource.RUSAGE_CHILDREN), time.time()
        r = subprocess.run([exe, "-t16", "-K", "10k", "--gff", mpi, faa], env=e, capture_output=True, timeout=900)
        r1, t1 = resource.getrusage(resource.RUSAGE_CHILDREN), time.time()
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        outs[devs] = r.stdout
        cpu[devs] = ((r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime), t1 - t0)
    assert outs["0,0,0,0,0,0,0,0"] == outs["0"]
    assert outs["0"].count(b"\n") > 40
    print("host CPU-s / wall-s: one pipeline %.2f / %.2f, eight pipelines %.2f / %.2f" % (cpu["0"] + cpu["0,0,0,0,0,0,0,0"]))
    ref = os.path.join(refbind.ROOT, "oracle", "_ref", "miniprot")
    if os.path.exists(ref):
        r = subprocess.run([ref, "-t4", "-K", "10k", "--gff", mpi, faa], capture_output=True, timeout=900)
        assert r.returncode == 0
        assert outs["0"] == r.stdout
