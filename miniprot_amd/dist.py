"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

miniprot's work is embarrassingly parallel over proteins (map.c:264-271): every rank maps its share of the job's
mini-batches against a replicated index and there is NO collective on the data path.  What the ranks exchange is small:

  1. (GFF3/GTF only) one all_gather of "how many hit records does each of my batches print", so that every rank can format
     its batches with the ids a single process would have given them -- the running hit counter of the output file
     (map.c:306, format.c:377,428) is a prefix sum over the batches in input order;
  2. the output text of every batch, sent point-to-point to rank 0 only (exact sizes, no padding), which writes the
     batches in input order.

Shards are whole mini-batches dealt round-robin (batch k -> rank k mod N): a batch is the unit the GPU pipeline is efficient
at, batches of one job are statistically alike (same protein-length law), and round-robin keeps every rank's pipeline fed
from the first batch on.  (SURVEY.md 8(e) suggests cost-balanced or stolen chunks; with i.i.d. batches a static deal is
balanced to within one batch, and it needs no shared queue.)
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def my_batches(n_batches, rank, world):
    """The mini-batches of a job that `rank` maps: k = rank, rank + world, ..."""
    return list(range(rank, n_batches, world))


def exchange_counts(mine, counts, n_batches, world, device):
    """all_gather of the per-batch output counts: returns int64[n_batches] on every rank."""
    import torch
    import torch.distributed as dist
    full = np.zeros(n_batches, dtype=np.int64)
    full[mine] = counts
    if world == 1 or not dist.is_initialized():
        return full
    per = (n_batches + world - 1) // world
    loc = torch.zeros(per, dtype=torch.int64, device=device)
    if len(mine):
        loc[:len(mine)] = torch.tensor(list(counts), dtype=torch.int64, device=device)
    parts = [torch.zeros(per, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, loc)
    for r in range(world):
        ks = my_batches(n_batches, r, world)
        full[ks] = parts[r][:len(ks)].cpu().numpy()
    return full


def gather_texts(mine, texts, n_batches, rank, world, device):
    """The text of every batch to rank 0, in batch order (rank 0: list of n_batches bytes objects; other ranks: None).
    One all_gather of the sizes, then the payload of each rank as ONE point-to-point message to rank 0."""
    import torch
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return list(texts)
    sizes = exchange_counts(mine, [len(t) for t in texts], n_batches, world, device)
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
        ks = my_batches(n_batches, r, world)
        total = int(sizes[ks].sum()) if len(ks) else 0
        if total:
            bufs[r] = torch.empty(total, dtype=torch.uint8, device=device)
            reqs.append(dist.irecv(bufs[r], src=r))
    for q in reqs:
        q.wait()
    for r, buf in bufs.items():
        blob = buf.cpu().numpy().tobytes()
        p = 0
        for k in my_batches(n_batches, r, world):
            out[k] = blob[p:p + int(sizes[k])]
            p += int(sizes[k])
    for k in range(n_batches):
        if out[k] is None:
            out[k] = b""
    return out


def run_sharded(n_batches, rank, world, device, needs_ids, map_mine, count_fn, format_fn):
    """One job = n_batches mini-batches over `world` ranks; rank 0 gets the output of a single-process run.

    map_mine(ks)         -> maps the batches ks (this rank's share) and returns one object per batch: the batch's text
                            (bytes) when needs_ids is False, else its structured result
    count_fn(k, res)     -> hit records the output of batch k prints (only with needs_ids)
    format_fn(k, res, id0) -> text of batch k whose first printed hit gets id0 + 1 (only with needs_ids)
    Returns (texts on rank 0 | None, ids consumed by the whole job)."""
    mine = my_batches(n_batches, rank, world)
    res = map_mine(mine)
    if not needs_ids:
        return gather_texts(mine, res, n_batches, rank, world, device), None
    counts = exchange_counts(mine, [count_fn(k, r) for k, r in zip(mine, res)], n_batches, world, device)
    id0 = np.concatenate([[0], np.cumsum(counts)])
    texts = [format_fn(k, r, int(id0[k])) for k, r in zip(mine, res)]
    return gather_texts(mine, texts, n_batches, rank, world, device), int(id0[-1])
