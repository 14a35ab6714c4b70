"""One rank of tests/test_hip_nccl_single_rank.py (launched by torch.distributed.run): the sharded batch pipeline on
the REAL WorldBatch backend with its small-result gather running as RCCL collectives on the device tensors."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_path):
    import torch
    import torch.distributed as dist

    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.distributed import ShardedWorldBatch, all_gather_ragged, max_over_ranks

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank, world = dist.get_rank(), dist.get_world_size()
    fs = 16000
    secs = [0.9, 0.35, 0.6, 0.5, 0.75]  # five ragged utterances
    xs = [synth_utterance(100 + i, fs, s) for i, s in enumerate(secs)]
    sb = ShardedWorldBatch(device_index=local_rank)
    res = {"backend": dist.get_backend(), "world": world, "rank": rank}
    for method, requiem in (("dio", False), ("harvest", True)):
        enc = sb.encode(xs, fs, f0_method=method, is_requiem=requiem)
        assert enc.f0.is_cuda
        dist.barrier()
        gathered = sb.gather_f0(use_collectives=True)  # all_gather_ragged on the device tensors: RCCL
        y, y_off = sb.decode(seed=7)
        # the unsharded result: a plain WorldBatch on the whole list
        wb = WorldBatch(local_rank)
        ref = wb.encode(xs, fs, f0_method=method, is_requiem=requiem)
        y_ref, off_ref = wb.decode_device(ref, seed=7)
        fo = ref.batch.frame_off
        f0_ref, vuv_ref = ref.f0.cpu().numpy(), ref.vuv.cpu().numpy()
        lo, hi = sb.range
        ok = len(gathered) == len(xs) if rank == 0 else gathered is None
        if rank == 0:
            for u in range(len(xs)):
                a, b = int(fo[u]), int(fo[u + 1])
                ok = ok and np.array_equal(gathered[u][0], f0_ref[a:b]) and np.array_equal(gathered[u][1], vuv_ref[a:b])
        # dense tensors stay on the rank: this rank's slice of the unsharded encoding, bitwise
        a, b = int(fo[lo]), int(fo[hi])
        ok = ok and torch.equal(enc.spectrogram, ref.spectrogram[a:b]) and torch.equal(enc.aperiodicity, ref.aperiodicity[a:b])
        ya, yb = int(off_ref[lo]), int(off_ref[hi])
        ok = ok and np.array_equal(np.asarray(y_off) + ya, np.asarray(off_ref[lo:hi + 1]))
        ok = ok and bool(torch.allclose(y, y_ref[ya:yb], atol=1e-13, rtol=0))
        # the typed ragged all-gather itself, on a device tensor
        parts = all_gather_ragged(enc.f0)
        ok = ok and len(parts) == world and parts[rank].is_cuda and torch.equal(parts[rank], enc.f0)
        res[method] = bool(ok)
    res["max_over_ranks"] = max_over_ranks(1.5 + rank, device=torch.device("cuda", local_rank))
    dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
