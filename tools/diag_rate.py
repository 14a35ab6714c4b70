"""Diagnostic (GPU): Harvest stage by stage against the oracle at one sampling rate: tools/diag_rate.py 11025"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import pitch_harvest
from world import _hip, _tables
from world._synthetic import synth_utterance
from world.harvest import harvest_device
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 11025
x = synth_utterance(90, fs, 0.6)
o = pitch_harvest.harvest_np(x, fs, return_aux=True)
aux = o["aux"]
print("aux keys", sorted(aux.keys()))
rt = _hip.Runtime.get()
nf = _tables.frame_count(len(x), fs, 5)
tp = _tables.frame_times(nf, 5)
batch = rt.make_batch([0, len(x)], [0, nf])
f0, vuv, dbg = harvest_device(rt, batch, rt.to_device(x), rt.to_device(tp), fs, debug=True)
print("flags", rt.take_flags())
y = dbg["y"].cpu().numpy()[: len(aux["y"])]
print("y max diff", np.max(np.abs(y - aux["y"])))
nb = aux["raw"].shape[0]
raw = dbg["raw"].cpu().numpy()[: nb * aux["raw"].shape[1]].reshape(nb, -1)
print("raw live mismatch", int(np.sum((raw != 0) != (aux["raw"] != 0))), "max diff", np.max(np.abs(raw - aux["raw"])))
f1 = dbg["f0_1ms"].cpu().numpy()[: len(aux["f0_1ms"])]
d = np.abs(f1 - aux["f0_1ms"])
print("f0_1ms voiced mismatch", int(np.sum((f1 != 0) != (aux["f0_1ms"] != 0))), "max diff", d.max(), "frames > 1e-6:", int((d > 1e-6).sum()), "of", len(d))
idx = np.argsort(d)[::-1][:8]
for i in idx:
    print("   1ms frame", i, "ours", f1[i], "oracle", aux["f0_1ms"][i])
