import os, sys
import numpy as np
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT,"python-world_amd")): sys.path.insert(0,p)
from oracle import pitch_swipe
from world._synthetic import synth_utterance
from world.swipe import swipe
for fs, floor in ((96000, 71), (88200, 71), (44100, 40.9), (48000, 45), (96000, 57.1), (88200, 50)):
    x = synth_utterance(7, fs, 0.5)
    o = pitch_swipe.swipe_np(fs, x, [floor, 800], sTHR=0.3)
    d = swipe(fs, x, [floor, 800], 0.005, 0.3)
    v = o["vuv"] > 0
    print(fs, floor, "vuv equal", np.array_equal(d["vuv"], o["vuv"]), "voiced", int(v.sum()), "f0 max rel", float(np.max(np.abs(d["f0"][v]-o["f0"][v])/o["f0"][v])) if v.any() else 0)
