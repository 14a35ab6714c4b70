"""The with_transfers_pipelined loop of bench.py on its own (kernel upload from pinned memory, DMA download on a private
stream, double-buffered), for `rocprofv3 --memory-copy-trace --kernel-trace`: the copy records show which direction
rode a DMA engine and that the downloads overlap the next step's kernels."""
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

fs = 16000
xs = bench.make_inputs(0, 64, fs, 10.0)
wb = WorldBatch(0)
batch, x_d, tp_d = wb.upload(xs, fs)
x_pin = torch.from_numpy(np.concatenate(xs)).pin_memory()


def one(k):
    wb.refill_from_pinned(x_d, x_pin)
    e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
    yy, _ = wb.decode_device(e, seed=10 + k, check=False)
    return wb.download_async((e.f0, e.vuv, e.spectrogram, e.aperiodicity, yy), slot=k % 2)


one(0), one(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(6):
    one(2 + k)
torch.cuda.synchronize()
print("pipelined ms/step %.2f" % ((time.perf_counter() - t0) / 6 * 1e3))
