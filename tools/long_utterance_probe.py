"""Probe (GPU): single very long utterances through the drop-in facade — 30 minutes at 16 kHz (Harvest + D4C-Requiem + Requiem
synthesis; DIO + D4C + pulse-wise synthesis) and 10 minutes at 48 kHz: no flag, finite output of the reference's length, the same
bits on a second run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)
from world import _hip
from world._synthetic import synth_utterance
from world.main import World
from world.synthesis import time_axis_params

w = World()
for fs, minutes, method, req in ((16000, 30, "harvest", True), (16000, 30, "dio", False), (48000, 10, "harvest", False)):
    base = synth_utterance(7, fs, 20.0)
    x = np.tile(base, int(minutes * 60 / 20))
    x = x * (1.0 + 0.1 * np.sin(np.arange(len(x)) * (2 * np.pi / (fs * 37.0))))  # (not periodic over the tiles)
    t = time.time()
    d = w.encode(fs, x, f0_method=method, is_requiem=req)
    t_enc = time.time() - t
    t = time.time()
    np.random.seed(1)
    y = w.decode(dict(d))["out"]
    t_dec = time.time() - t
    ny = time_axis_params(d["temporal_positions"], fs)[0]
    d2 = w.encode(fs, x, f0_method=method, is_requiem=req)
    same = all(np.array_equal(d[k], d2[k]) for k in ("f0", "vuv", "spectrogram", "aperiodicity"))
    print("%d min at %d Hz, %s%s: %d frames (%d voiced) in %.2f s, decode %d samples (expected %d) in %.2f s, finite %s, second run identical %s, flags %s" % (
        minutes, fs, method, " + Requiem" if req else "", len(d["f0"]), int(d["vuv"].sum()), t_enc, len(y), ny, t_dec,
        bool(np.isfinite(y).all() and np.isfinite(d["spectrogram"]).all()), same, [i for i, f in enumerate(_hip.Runtime.get().take_flags()) if f]), flush=True)
print("PROBE DONE")
