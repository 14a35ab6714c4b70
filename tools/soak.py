#!/usr/bin/env python3
"""Soak (GPU): the drop-in facade in a loop on batches of changing shape — host RSS, device memory in use (hipMemGetInfo) and
torch's own pool are sampled every iteration; a leak shows as a slope.  tools/soak.py [iterations] [seconds budget]"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    from world._synthetic import synth_utterance
    from world.main import World

    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    rng = np.random.RandomState(3)
    pool = {fs: [synth_utterance(u, fs, 3.0) for u in range(12)] for fs in (16000, 22050, 48000)}
    w = World()
    rows = []
    t0 = time.time()
    for it in range(iters):
        fs = int(rng.choice([16000, 16000, 22050, 48000]))
        n = int(rng.randint(1, 13))
        xs = [pool[fs][int(rng.randint(12))][: int(fs * (0.2 + 2.8 * rng.rand()))] for _ in range(n)]
        method = "harvest" if rng.rand() < 0.5 else "dio"
        req = bool(rng.rand() < 0.5)
        devs = [0, 0] if (it % 4 == 1 and n >= 2) else None  # the thread-per-device pool (two contexts on the one GPU) now and then
        dats = w.encode_batch(fs, xs, f0_method=method, is_requiem=req, devices=devs)
        if rng.rand() < 0.5:
            for d in dats:
                w.scale_pitch(d, 0.7 + rng.rand())
        if rng.rand() < 0.3:
            _ = dats[0]["spectrogram"]  # (materialise one dense tensor now and then)
        outs = w.decode_batch(dats, devices=devs)
        assert all(np.isfinite(d["out"]).all() for d in outs)
        if it % 3 == 0:  # the single-utterance facade as well
            d1 = w.encode(fs, xs[0], f0_method=method, is_requiem=req)
            w.decode(d1)
        del dats, outs
        free, total = torch.cuda.mem_get_info()
        rows.append((it, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, (total - free) / 2 ** 20,
                     torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20))
        if time.time() - t0 > budget:
            break
    a = np.array(rows)
    k = len(a)
    third = max(1, k // 3)
    def slope(col):
        return float(np.polyfit(a[third:, 0], a[third:, col], 1)[0])
    print("SOAK iterations %d in %.0f s" % (k, time.time() - t0))
    for name, col in (("host max RSS MB", 1), ("device memory in use MB", 2), ("torch allocated MB", 3), ("torch reserved MB", 4)):
        print("  %-24s first third max %.0f, last third max %.0f, slope over the last two thirds %.3f MB/iteration" % (
            name, a[:third, col].max(), a[-third:, col].max(), slope(col)))


if __name__ == "__main__":
    main()
