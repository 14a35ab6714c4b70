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


class _StubEncoding:
    """What ShardedWorldBatch needs from an encoding, on the CPU: per-utterance frame offsets and flat f0 / vuv."""

    def __init__(self, xs, fs, first_marker):
        import torch

        nfs = [int(1000 * len(x) / fs / 5 + 1) for x in xs]
        self.n_utt = len(xs)
        self.batch = type("B", (), {"frame_off": np.concatenate([[0], np.cumsum(nfs)])})()
        # the stub "analysis": f0 of utterance = its first sample (a marker the parent can check), vuv = 1
        self.f0 = torch.cat([torch.full((n,), float(x[0]), dtype=torch.float64) for n, x in zip(nfs, xs)])
        self.vuv = torch.ones(int(sum(nfs)), dtype=torch.float64)
        self.lens = [len(x) for x in xs]


class _StubBackend:
    def encode(self, xs, fs, **kw):
        return _StubEncoding(xs, fs, None)

    def decode_device(self, enc, **kw):
        import torch

        y_off = np.concatenate([[0], np.cumsum(enc.lens)])
        return torch.zeros(int(y_off[-1]), dtype=torch.float64), y_off


def _sharded_worker(rank, world, port, lengths, ret):
    import torch.distributed as dist

    from world.distributed import ShardedWorldBatch, shard_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = ShardedWorldBatch(backend=_StubBackend())
    made = []

    def loader(u):  # a rank only materialises its own utterances
        made.append(u)
        return np.full(lengths[u], float(u))

    enc = sb.encode(loader, 16000, lengths=lengths)
    lo, hi = shard_ranges(lengths, world)[rank]
    ok = sb.range == (lo, hi) and made == list(range(lo, hi)) and enc.n_utt == hi - lo
    y, y_off = sb.decode()
    ok = ok and int(y_off[-1]) == sum(lengths[lo:hi])
    got = sb.gather_f0(dst=0)
    if rank == 0:
        ret["n"] = len(got)
        ret["order_ok"] = all(np.all(f0 == float(u)) and len(f0) == int(1000 * lengths[u] / 16000 / 5 + 1)
                              for u, (f0, vuv) in enumerate(got))
    else:
        ok = ok and got is None
    ret["ok%d" % rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_world_batch():
    """The product's sharded batch driver (world.distributed.ShardedWorldBatch) with a stubbed per-rank compute:
    each rank loads exactly its shard_ranges slice, decodes it, and rank 0 receives every utterance's f0 in order."""
    import torch.multiprocessing as mp

    lengths = [160000, 80000, 120000, 160000, 40000, 90000, 160000]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(2, _free_port(), lengths, ret), nprocs=2, join=True)
    assert ret["ok0"] and ret["ok1"] and ret["n"] == len(lengths) and ret["order_ok"]


def _ragged_worker(rank, world, port, lengths, ret):
    import torch
    import torch.distributed as dist

    from world.distributed import ShardedWorldBatch, all_gather_ragged, shard_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # typed ragged all-gather on its own: rank r contributes r*3 values (rank 0: none)
    parts = all_gather_ragged(torch.arange(rank * 3, dtype=torch.float64) + 100 * rank)
    ok = [p.tolist() for p in parts] == [[100.0 * r + i for i in range(3 * r)] for r in range(world)]
    sb = ShardedWorldBatch(backend=_StubBackend())
    enc = sb.encode(lambda u: np.full(lengths[u], float(u)), 16000, lengths=lengths)
    lo, hi = shard_ranges(lengths, world)[rank]
    ok = ok and sb.range == (lo, hi) and ((enc is None) == (lo == hi))
    ok = ok and ((sb.decode() is None) == (lo == hi))
    got = sb.gather_f0(dst=0)
    if rank == 0:
        ret["n"] = len(got)
        ret["order_ok"] = all(np.all(f0 == float(u)) and np.all(vuv == 1.0)
                              and len(f0) == len(vuv) == int(1000 * lengths[u] / 16000 / 5 + 1)
                              for u, (f0, vuv) in enumerate(got))
    else:
        ok = ok and got is None
    ret["ok%d" % rank] = bool(ok)
    ret["empty%d" % rank] = lo == hi
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_uneven_lengths_and_an_empty_shard():
    """world_size 4 with fewer utterances than ranks: one rank gets an empty shard and still takes part in the typed
    collectives; the others hold uneven frame counts; rank 0 receives every contour in utterance order."""
    import torch.multiprocessing as mp

    lengths = [160000, 30000, 95000]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ragged_worker, args=(4, _free_port(), lengths, ret), nprocs=4, join=True)
    assert all(ret["ok%d" % r] for r in range(4))
    assert sum(ret["empty%d" % r] for r in range(4)) == 1
    assert ret["n"] == len(lengths) and ret["order_ok"]


def test_four_ranks_many_uneven_utterances():
    import torch.multiprocessing as mp

    lengths = [160000, 80000, 120000, 160000, 40000, 90000, 160000, 20000, 155000, 64000, 31000]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ragged_worker, args=(4, _free_port(), lengths, ret), nprocs=4, join=True)
    assert all(ret["ok%d" % r] for r in range(4)) and not any(ret["empty%d" % r] for r in range(4))
    assert ret["n"] == len(lengths) and ret["order_ok"]


def test_sharded_world_batch_single_process():
    from world.distributed import ShardedWorldBatch

    sb = ShardedWorldBatch(backend=_StubBackend())
    xs = [np.full(8000 * (i + 1), float(i)) for i in range(3)]
    enc = sb.encode(xs, 16000)
    assert sb.range == (0, 3) and enc.n_utt == 3
    got = sb.gather_f0()
    assert [float(f0[0]) for f0, _ in got] == [0.0, 1.0, 2.0]


def test_shard_ranges_for_the_benchmark_configs():
    """BASELINE.json configs 4 and 5 over the 8 GPUs of a node: 1024 x 10 s at 16 kHz -> 128 utterances per rank,
    128 x 60 s at 48 kHz -> 16 per rank; ranges contiguous, in rank order, covering the batch once."""
    from world.distributed import shard_ranges

    for n_utt, samples, world, per in ((1024, 160000, 8, 128), (128, 2880000, 8, 16), (1024, 160000, 4, 256),
                                       (1024, 160000, 2, 512), (64, 160000, 8, 8)):
        r = shard_ranges([samples] * n_utt, world)
        assert [b - a for a, b in r] == [per] * world
        assert r[0][0] == 0 and r[-1][1] == n_utt and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    # ragged: balanced by samples, not by count
    lengths = [160000] * 8 + [16000] * 80
    r = shard_ranges(lengths, 2)
    tot = [sum(lengths[a:b]) for a, b in r]
    assert abs(tot[0] - tot[1]) <= 160000 and r[0][1] == r[1][0]
