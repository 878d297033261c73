"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).

miniprot's work is embarrassingly parallel over proteins (map.c:264-271): every rank maps its own shard
against a replicated index and there is NO collective on the data path.  The only exchange is the final
gather of the (small) output to rank 0, done here with one all_gather of sizes and one of padded bytes."""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_bytes(payload, rank, world, device):
    """Gather one bytes object per rank to rank 0 (returns the list on rank 0, None elsewhere)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not dist.is_initialized():
        return [payload]
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = max(int(s.item()) for s in sizes)
    buf = torch.zeros(max(m, 1), dtype=torch.uint8, device=device)
    if len(payload):
        buf[:len(payload)] = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy()).to(device)
    parts = [torch.zeros(max(m, 1), dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(parts, buf)
    if rank != 0:
        return None
    return [bytes(p[:int(s.item())].cpu().numpy()) for p, s in zip(parts, sizes)]
