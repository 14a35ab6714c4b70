"""with_transfers_pipelined variants (bench.py): upload by DMA copy or by a mapped-memory kernel, download by DMA copy
or by a mapped-memory kernel; each variant timed `reps` times in this process.  python tools/pipe_variants.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from world.batch import WorldBatch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fs = 16000
xs = bench.make_inputs(0, 64, fs, 10.0)
wb = WorldBatch(0)
batch, x_d, tp_d = wb.upload(xs, fs)
x_pin = torch.from_numpy(np.concatenate(xs)).pin_memory()


def run(up_mapped, down_mapped, steps=6):
    def one(k):
        if up_mapped:
            wb.refill_from_pinned(x_d, x_pin)
        else:
            x_d.copy_(x_pin, non_blocking=True)
        e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
        yy, _ = wb.decode_device(e, seed=10 + k, check=False)
        return wb.download_async((e.f0, e.vuv, e.spectrogram, e.aperiodicity, yy), slot=k % 2, mapped=down_mapped)

    one(0)
    one(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(2 + k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for up in (False, True):
    for down in (False, True):
        print("upload %-6s download %-6s ms/step:" % ("kernel" if up else "dma", "kernel" if down else "dma"),
              " ".join("%.1f" % run(up, down) for _ in range(reps)), flush=True)
wb.check("pipe_variants")
