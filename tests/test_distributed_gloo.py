"""world_size-2 gloo test of the multi-GPU plumbing (runs on CPU): sharding, gather of small results to
rank 0 in utterance order, barrier + max-over-ranks timing.  The per-rank compute itself needs a GPU and
is covered by the -m gpu tests; nothing here launches a kernel."""
import os
import socket

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, ret):
    import torch.distributed as dist

    from world.distributed import gather_small, max_over_ranks, my_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = my_range(lengths, world, rank)
    # stand-in for the per-utterance small outputs (f0 contour): deterministic function of the utterance id
    local = [np.full(int(1000 * lengths[u] / 16000 / 5 + 1), float(u)) for u in range(s, e)]
    dist.barrier()
    got = gather_small(local, dst=0)
    t = max_over_ranks(0.25 + rank)
    if rank == 0:
        ret["n"] = len(got)
        ret["order_ok"] = all(np.all(a == float(i)) for i, a in enumerate(got))
        ret["frames_ok"] = all(len(a) == int(1000 * lengths[i] / 16000 / 5 + 1) for i, a in enumerate(got))
    ret["t%d" % rank] = t
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_and_timing():
    import torch.multiprocessing as mp

    lengths = [160000, 80000, 120000, 160000, 40000]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, lengths, ret), nprocs=2, join=True)
    assert ret["n"] == len(lengths) and ret["order_ok"] and ret["frames_ok"]
    assert ret["t0"] == ret["t1"] == 1.25
