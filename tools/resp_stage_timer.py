"""Per-stage cycle counts inside response_kernel (build with WH_EXTRA_FLAGS=-DWH_RESP_STAGE_TIMER)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch
from world._synthetic import synth_utterance
from world.batch import WorldBatch

# tools/resp_stage_timer.py [fs] [utterances] [seconds] [pitch scale] [duration scale]   (config 5: 48000 16 60 1.5 2.0)
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
wb = WorldBatch(0)
xs = [synth_utterance(i, fs, secs) for i in range(n_utt)]
batch, x_d, tp_d = wb.upload(xs, fs)
enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio" if fs <= 16000 else "harvest")
if len(sys.argv) > 4:
    enc.scale_pitch(float(sys.argv[4]))
if len(sys.argv) > 5:
    enc.scale_duration(float(sys.argv[5]))
lib = wb.rt.lib
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    y, _ = wb.decode_device(enc, seed=it)
    torch.cuda.synchronize()
    lib.wh_debug_resp_stages(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = ["setup: pulse look-up, 4 spectral rows, noise, mean", "2 x (log, rFFT, fold, rFFT, exp) + fractional delay",
         "2 x inverse real FFT", "response reorder + noise convolution", "DC sum + overlap-add atomics"]
tot = v[:5].sum()
for n, c in zip(names, v[:5]):
    print("%-55s %6.1f %%" % (n, 100 * c / tot))
print("total cycles (sum over pulses) %.3e" % tot)
