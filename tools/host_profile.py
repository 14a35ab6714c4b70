"""Host-side cost of enqueueing the config-2 step (cProfile over a few asynchronous steps)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch
from world._synthetic import synth_utterance
from world.batch import WorldBatchLanes

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = WorldBatchLanes(0, lanes=lanes)
xs = [synth_utterance(i, 16000, 10.0) for i in range(64)]
wl.upload(xs, 16000)


def step(seed):
    encs = wl.encode_device(16000, f0_method="dio")
    return wl.decode_device(encs, seed=seed)


for w in range(2):
    step(w)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(5):
    step(10 + k)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
