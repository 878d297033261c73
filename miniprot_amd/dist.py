"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

miniprot's work is embarrassingly parallel over proteins (map.c:264-271): every rank maps its share of the job's
mini-batches against a replicated index and there is NO collective on the data path.  What the ranks exchange is small:

  1. which batches each rank mapped and (GFF3/GTF only) how many hit records each of them prints, by all_gather, so that
     every rank can format its batches with the ids a single process would have given them -- the running hit counter of the
     output file (map.c:306, format.c:377,428) is a prefix sum over the batches in input order;
  2. the output text of every batch, sent point-to-point to rank 0 only (exact sizes, no padding), which writes the
     batches in input order.

Dealing.  The reference deals queries to its threads by work stealing (kt_for, map.c:264-271): the cost of a
protein is not known before it has been seeded.  Between processes the unit is a mini-batch -- what the GPU pipeline is
efficient at -- and the deal is DYNAMIC by default: the ranks share one counter (an atomic add in the rendezvous store of
torch.distributed) and a rank's pipeline claims the next batch of the job whenever it has room for one
(mpa_map_batches_claim), so a rank that gets through its batches faster maps more of them and no rank idles while another
still has a queue.  `my_batches` (round-robin, batch k -> rank k mod N) is the static deal, kept for callers without a
store and for the tests.
"""
import numpy as np

_job_serial = [0]


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def my_batches(n_batches, rank, world):
    """The mini-batches of a job that `rank` maps under the static deal: k = rank, rank + world, ..."""
    return list(range(rank, n_batches, world))


def job_counter(n_batches, world):
    """claim() for one job: returns the next unclaimed batch index of the job, -1 when there is none left.  All ranks must call
    job_counter() for the same jobs in the same order (the key of the shared counter is a per-process serial number)."""
    import torch.distributed as dist
    _job_serial[0] += 1
    if world == 1 or not dist.is_initialized():
        state = [0]
        def claim_local():
            k = state[0]
            state[0] += 1
            return k if k < n_batches else -1
        return claim_local
    store = dist.distributed_c10d._get_default_store()
    key = "mpa_job_%d" % _job_serial[0]
    def claim():
        k = store.add(key, 1) - 1                              # atomic fetch-and-add on the rendezvous store (rank 0's server)
        return k if k < n_batches else -1
    return claim


def any_rank_failed(failed, world, device):
    """True on every rank iff `failed` is true on at least one (one small all_reduce; no-op for a single process)."""
    import torch
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return bool(failed)
    flag = torch.tensor([1 if failed else 0], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(int(flag.item()))


def exchange_values(mine, values, n_batches, world, device):
    """Every rank contributes (batch index, value) for the batches it mapped, in the order it mapped them.  Returns
    (int64[n_batches] of the values, [batch list of rank 0, of rank 1, ...]) on every rank."""
    import torch
    import torch.distributed as dist
    full = np.zeros(n_batches, dtype=np.int64)
    if len(mine):
        full[list(mine)] = list(values)
    if world == 1 or not dist.is_initialized():
        return full, [list(mine)]
    cnt = torch.tensor([len(mine)], dtype=torch.int64, device=device)
    cnts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(c.item()) for c in cnts]
    per = max(max(cnts), 1)
    loc = torch.zeros((per, 2), dtype=torch.int64, device=device)
    if len(mine):
        loc[:len(mine), 0] = torch.tensor(list(mine), dtype=torch.int64, device=device)
        loc[:len(mine), 1] = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    parts = [torch.zeros((per, 2), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, loc)
    lists = []
    for r in range(world):
        p = parts[r][:cnts[r]].cpu().numpy()
        ks = [int(x) for x in p[:, 0]]
        full[ks] = p[:, 1]
        lists.append(ks)
    return full, lists


def exchange_counts(mine, counts, n_batches, world, device):
    """all_gather of the per-batch output counts: returns int64[n_batches] on every rank."""
    return exchange_values(mine, counts, n_batches, world, device)[0]


def gather_texts(mine, texts, n_batches, rank, world, device):
    """The text of every batch to rank 0, in batch order (rank 0: list of n_batches bytes objects; other ranks: None).
    One all_gather of (batch, size), then the payload of each rank as ONE point-to-point message to rank 0."""
    import torch
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        out = [b""] * n_batches
        for k, t in zip(mine, texts):
            out[k] = t
        return out
    sizes, lists = exchange_values(mine, [len(t) for t in texts], n_batches, world, device)
    if rank != 0:
        blob = b"".join(texts)
        if len(blob):
            dist.send(torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(device), dst=0)
        return None
    out = [None] * n_batches
    for k, t in zip(mine, texts):
        out[k] = t
    bufs, reqs = {}, []
    for r in range(1, world):
        total = int(sizes[lists[r]].sum()) if len(lists[r]) else 0
        if total:
            bufs[r] = torch.empty(total, dtype=torch.uint8, device=device)
            reqs.append(dist.irecv(bufs[r], src=r))
    for q in reqs:
        q.wait()
    for r, buf in bufs.items():
        blob = buf.cpu().numpy().tobytes()
        p = 0
        for k in lists[r]:
            out[k] = blob[p:p + int(sizes[k])]
            p += int(sizes[k])
    for k in range(n_batches):
        if out[k] is None:
            out[k] = b""
    return out


def run_sharded(n_batches, rank, world, device, needs_ids, map_mine, count_fn, format_fn, map_claimed=None):
    """One job = n_batches mini-batches over `world` ranks; rank 0 gets the output of a single-process run.

    map_mine(ks)         -> maps the batches ks (this rank's share under the static deal) and returns one object per batch: the
                            batch's text (bytes) when needs_ids is False, else its structured result
    map_claimed(claim)   -> (dynamic deal, used when given) maps the batches claim() hands out until it returns -1 and returns
                            (ks, objects) for the batches it got, in the order it got them
    count_fn(k, res)     -> hit records the output of batch k prints (only with needs_ids)
    format_fn(k, res, id0) -> text of batch k whose first printed hit gets id0 + 1 (only with needs_ids)
    Returns (texts on rank 0 | None, ids consumed by the whole job)."""
    err = None
    try:
        if map_claimed is not None:
            mine, res = map_claimed(job_counter(n_batches, world))
            mine = list(mine)
        else:
            mine = my_batches(n_batches, rank, world)
            res = map_mine(mine)
    except Exception as e:                                       # (a failed claim callback, a mapping error: this rank has nothing to gather)
        err, mine, res = e, [], []
    # every rank learns whether ANY rank failed before the first collective of the gather: a rank that raised alone would leave
    # the others waiting in all_gather for ever
    if any_rank_failed(err is not None, world, device):
        if err is not None and (world == 1 or rank is None):
            raise err                                            # (a single process: the caller gets the error it caused, type and traceback)
        if err is not None:
            raise RuntimeError("run_sharded: mapping failed on this rank: %r" % (err,)) from err
        raise RuntimeError("run_sharded: mapping failed on another rank")
    if not needs_ids:
        return gather_texts(mine, res, n_batches, rank, world, device), None
    counts = exchange_counts(mine, [count_fn(k, r) for k, r in zip(mine, res)], n_batches, world, device)
    id0 = np.concatenate([[0], np.cumsum(counts)])
    texts = [format_fn(k, r, int(id0[k])) for k, r in zip(mine, res)]
    return gather_texts(mine, texts, n_batches, rank, world, device), int(id0[-1])
