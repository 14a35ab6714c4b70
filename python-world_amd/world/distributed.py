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


def all_gather_ragged(t, group=None):
    """Typed all-gather of one 1-D tensor per rank with a different length on every rank (possibly 0): the lengths
    travel first (one int64 each), then the tensors padded to the longest — two fixed-size ``all_gather`` collectives
    on the tensors' own device (RCCL over xGMI for device tensors, gloo for CPU ones), no pickling, no host staging.
    Returns the list of per-rank tensors, in rank order, on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    width = max(sizes)
    if width == 0:
        return [t.new_zeros((0,)) for _ in range(world)]
    padded = t.new_zeros((width,))
    padded[:t.numel()] = t.reshape(-1)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return [p[:k] for p, k in zip(parts, sizes)]


def collective_device(fallback, group=None):
    """Device the process group's collectives run on: the caller's GPU under "nccl" (= RCCL), the CPU under gloo."""
    import torch
    import torch.distributed as dist

    return fallback if dist.get_backend(group) == "nccl" else torch.device("cpu")


class ShardedWorldBatch:
    """One 1024-utterance (or any) batch over the GPUs of a node: rank r encodes / decodes the contiguous utterance
    range shard_ranges gives it on its own GPU, with no collective on the data path; the dense
    outputs stay on the rank that produced them, the small per-utterance results (f0, vuv, Requiem band
    aperiodicity, output lengths) can be gathered to one rank.  One process per GPU; works un-initialised
    (world_size 1) as a plain single-GPU batch.

    ``backend``: object with ``encode(xs, fs, **kw) -> enc`` and ``decode_device(enc, **kw) -> (y, y_off)``; default
    ``world.batch.WorldBatch`` on ``device_index`` (LOCAL_RANK).  The CPU (gloo) test passes a stub here."""

    def __init__(self, device_index=None, backend=None, group=None):
        import os

        self.group = group
        self.world, self.rank = 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        except ImportError:
            pass
        if backend is None:
            from .batch import WorldBatch
            if device_index is None:
                device_index = int(os.environ.get("LOCAL_RANK", "0"))
            backend = WorldBatch(device_index)
        self.backend = backend
        self.range = (0, 0)
        self.lengths = None
        self.enc = None  # this rank's encoding once encode() has run (decode / gather_f0 before that: None)

    def shard(self, lengths):
        """[start, end) of this rank for a batch with the given utterance lengths (samples)."""
        self.lengths = [int(n) for n in lengths]
        self.range = my_range(self.lengths, self.world, self.rank)
        return self.range

    def encode(self, utterances, fs, lengths=None, **kw):
        """``utterances``: the whole batch as a list of 1-D arrays — every rank passes the same list and takes its
        slice — or a callable ``u -> waveform`` together with ``lengths`` so that a rank only ever materialises its own
        utterances.  Returns this rank's encoding (None for an empty shard)."""
        if callable(utterances):
            lo, hi = self.shard(lengths)
            mine = [utterances(u) for u in range(lo, hi)]
        else:
            lo, hi = self.shard([len(x) for x in utterances])
            mine = list(utterances[lo:hi])
        self.enc = self.backend.encode(mine, fs, **kw) if mine else None
        return self.enc

    def decode(self, enc=None, **kw):
        """This rank's (y, y_off), or None for an empty shard."""
        enc = self.enc if enc is None else enc
        return self.backend.decode_device(enc, **kw) if enc is not None else None

    def gather_small(self, per_utterance, dst=0):
        """Gather a list with one small NumPy array per LOCAL utterance to ``dst`` in global utterance order."""
        if self.world == 1:
            return list(per_utterance)
        return gather_small(list(per_utterance), group=self.group, dst=dst)

    def gather_f0(self, enc=None, dst=0, use_collectives=None):
        """[(f0, vuv)] per utterance of the whole batch on ``dst`` (None elsewhere): <= 16 B per frame over xGMI.
        Typed collectives on the device tensors themselves (``all_gather_ragged``): the frame counts, then f0 and vuv
        stacked — the contours never pass through pickle or a host buffer on their way between GPUs.
        ``use_collectives``: default = only when there is more than one rank; True runs the collectives also in a
        one-rank group (the single-GPU rehearsal of the RCCL path, tests/test_hip_nccl_single_rank.py)."""
        enc = self.enc if enc is None else enc
        if use_collectives is None:
            use_collectives = self.world > 1
        if not use_collectives:
            if enc is None:
                return []
            fo = enc.batch.frame_off
            f0, vuv = enc.f0.cpu().numpy(), enc.vuv.cpu().numpy()
            return [(f0[int(fo[u]):int(fo[u + 1])].copy(), vuv[int(fo[u]):int(fo[u + 1])].copy())
                    for u in range(enc.n_utt)]
        import torch

        if enc is not None:
            dev = collective_device(enc.f0.device, self.group)
            counts = torch.as_tensor(np.diff(np.asarray(enc.batch.frame_off)), dtype=torch.int64, device=dev)
            both = torch.cat([enc.f0.reshape(-1).to(dev), enc.vuv.reshape(-1).to(dev)])
        else:  # an empty shard still takes part in the collectives
            dev = collective_device(torch.device("cuda", torch.cuda.current_device())
                                    if torch.cuda.is_available() else torch.device("cpu"), self.group)
            counts = torch.zeros((0,), dtype=torch.int64, device=dev)
            both = torch.zeros((0,), dtype=torch.float64, device=dev)
        all_counts = all_gather_ragged(counts, self.group)
        all_both = all_gather_ragged(both, self.group)
        if self.rank != dst:
            return None
        out = []
        for cnt, vals in zip(all_counts, all_both):
            cnt = cnt.cpu().numpy()
            vals = vals.cpu().numpy()
            half = len(vals) // 2
            off = np.concatenate([[0], np.cumsum(cnt)])
            for u in range(len(cnt)):
                out.append((vals[int(off[u]):int(off[u + 1])].copy(), vals[half + int(off[u]):half + int(off[u + 1])].copy()))
        return out
