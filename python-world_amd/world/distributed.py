"""Multi-GPU sharding of utterance batches.

Utterances are independent, so a batch is split into contiguous per-rank ranges balanced by sample
count and every rank runs the whole pipeline on its range: there is NO collective on the data path.
The only communication is an optional gather of the small per-utterance results (f0 / vuv / band
aperiodicity) to rank 0 and the barrier / max-reduce used for timing.  One process per GPU,
torch.distributed ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import numpy as np


def shard_ranges(lengths, world_size):
    """Contiguous [start, end) utterance ranges, one per rank, balanced by total samples.

    Greedy prefix split at the ideal cumulative boundaries; every rank gets at least one utterance while
    there are enough utterances, ranges are in rank order and cover the batch exactly once."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    csum = np.concatenate([[0], np.cumsum(lengths)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        cut = int(np.searchsorted(csum, target, side="left"))
        # pick the nearer prefix boundary, keep ranges non-empty and monotone
        if cut > 0 and abs(csum[cut - 1] - target) <= abs(csum[min(cut, n)] - target):
            cut -= 1
        lo = bounds[-1] + (1 if n - bounds[-1] > world_size - r else 0)
        hi = n - (world_size - r) if n >= world_size else n
        cut = max(min(cut, hi), min(lo, n))
        bounds.append(cut)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def my_range(lengths, world_size, rank):
    return shard_ranges(lengths, world_size)[rank]


def gather_small(local_items, group=None, dst=0):
    """Gather a list of small per-utterance NumPy arrays (e.g. f0 contours) from every rank to ``dst`` in
    utterance order.  Returns the concatenated list on ``dst`` and None elsewhere."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local_items, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


def max_over_ranks(value, device=None, group=None):
    """Max-reduce a Python float over all ranks (timing)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
