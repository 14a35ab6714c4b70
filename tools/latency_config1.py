"""Single-call latency of the drop-in on the reference's own benchmark (test/speed.py:13-18 of the reference: ONE
World().encode(fs, x, f0_method='harvest') on test-mwm.wav), cold and warm, plus decode, with a cProfile of the warm
encode.  Run on the GPU box: python tools/latency_config1.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import numpy as np
import torch
from scipy.io import wavfile

from world import main

fs, xi = wavfile.read(os.path.join(ROOT, "tests", "golden", "test-mwm.wav"))
x = xi / (2 ** 15 - 1)
W = main.World()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def timed(fn):
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


dat, cold = timed(lambda: W.encode(fs, x, f0_method="harvest"))
_, cold_dec = timed(lambda: W.decode(dict(dat)))
enc_ms, dec_ms = [], []
for k in range(7):
    dat, t = timed(lambda: W.encode(fs, x, f0_method="harvest"))
    enc_ms.append(t)
    _, t = timed(lambda: W.decode(dict(dat)))
    dec_ms.append(t)
print("encode cold %.1f ms, warm median %.2f ms (%s)" % (cold, np.median(enc_ms), ", ".join("%.1f" % v for v in enc_ms)))
print("decode cold %.1f ms, warm median %.2f ms (%s)" % (cold_dec, np.median(dec_ms), ", ".join("%.1f" % v for v in dec_ms)))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for k in range(5):
        dat = W.encode(fs, x, f0_method="harvest")
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    pr = cProfile.Profile()
    pr.enable()
    for k in range(5):
        W.decode(dict(dat))
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
