"""Per-stage cycle counts inside d4c_kernel (build a variant with wh_d4c:-DWH_D4C_STAGE_TIMER and point WH_LIB at it):
workgroup-thread-0 timestamps at the stage boundaries, summed over all voiced frames of config 2's batch (or, with the
argument 48000, of a 48 kHz batch: the N = 4096 instance)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch
from world._synthetic import synth_utterance
from world.batch import WorldBatch

FS = int(sys.argv[1]) if len(sys.argv) > 1 else 16000  # 48000: the N = 4096 instance (five band stages)
wb = WorldBatch(0)
xs = [synth_utterance(i, FS, 10.0) for i in range(64 if FS == 16000 else 16)]
batch, x_d, tp_d = wb.upload(xs, FS)
lib = wb.rt.lib
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    enc = wb.encode_device(batch, x_d, tp_d, FS, f0_method="dio")
    torch.cuda.synchronize()
    lib.wh_debug_d4c_stages(buf, 1)
v = np.array(list(buf), dtype=np.float64)
order = [(7, "start-up + the two stage-1 windows"), (8, "fused gate/power FFT"), (0, "gate reduction + power fold"),
         (1, "centroid A (window + FFT + fold)"), (2, "centroid B"), (3, "low-band replica (cent)"),
         (4, "smoothing: power replica + 3 sliding windows"), (12, "band: shaped group delay x Nuttall window -> buffer"), (13, "band: real FFT"), (9, "band: power"), (5, "rank select"),
         (6, "outputs"), (10, "  (inside the 4 windows: set-up + first walk)"), (11, "  (inside the 4 windows: reduction)")]
tot = v.sum()
voiced = float((enc.vuv.cpu().numpy() != 0).sum())
for i, n in order:
    print("%-48s %6.1f %%   %8.0f cycles / voiced frame" % (n, 100 * v[i] / tot, v[i] / max(voiced, 1)))
print("total %.0f cycles per voiced frame (workgroup latency)" % (tot / max(voiced, 1)))
