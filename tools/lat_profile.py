"""Host-side profile of the single-utterance facade (the reference's own benchmark call): World().encode / decode of the
test recording, cProfile of the warm calls.  usage: python tools/lat_profile.py [harvest|dio] [requiem]"""
import cProfile
import os
import pstats
import sys
import time
import wave

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "python-world_amd"))
from world.main import World  # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "harvest"
requiem = len(sys.argv) > 2 and sys.argv[2] == "requiem"
w = wave.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "test-mwm.wav"))
fs = w.getframerate()
x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float64) / (2 ** 15 - 1)
voc = World()
for _ in range(3):
    dat = voc.encode(fs, x, f0_method=method, is_requiem=requiem)
    voc.decode(dat)
import torch  # noqa: E402

for name, fn in (("encode", lambda: voc.encode(fs, x, f0_method=method, is_requiem=requiem)), ("decode", lambda: voc.decode(dat))):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    print(name, "ms:", " ".join("%.2f" % v for v in ts))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        fn()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(18)
