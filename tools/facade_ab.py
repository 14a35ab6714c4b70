"""A/B of the facade's resynthesis flow on the GPU box: copy_out on / off (World.decode_batch), 64 x 10 s, pitch x 1.5,
duration x 2.  usage: python tools/facade_ab.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch
from world import main
from world._synthetic import synth_utterance
fs = 16000
xs = [synth_utterance(u, fs, 10.0) for u in range(64)]
W = main.World()
def flow(copy_out):
    dats = W.encode_batch(fs, xs, f0_method="dio")
    for d in dats:
        W.scale_pitch(d, 1.5); W.scale_duration(d, 2.0)
    return W.decode_batch(dats, copy_out=copy_out)
for co in (False, True, False, True):
    for _ in range(3): flow(co)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); flow(co); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("copy_out", co, "median ms %.2f  min %.2f  max %.2f" % (np.median(ts) * 1e3, min(ts) * 1e3, max(ts) * 1e3))
