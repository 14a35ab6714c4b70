"""Diagnostic (GPU): one case of tools/differential_campaign.py, Harvest stage by stage against the oracle.
    python tools/campaign_case_diag.py <seed> <case index>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import differential_campaign as dc
from oracle import pitch_harvest
from world import _hip, _tables
from world.harvest import harvest_device, counted_event_caps, flat_samples

seed, idx = int(sys.argv[1]), int(sys.argv[2])
c = dc.draw_case(idx, seed)
print(c)
x = dc.make_input(c)
fs = c["fs"]
kw = c["kw"]
args = (kw.get("f0_floor", 71), kw.get("f0_ceil", 800), kw.get("frame_period", 5))
o = pitch_harvest.harvest_np(x, fs, *args, return_aux=True)
aux = o["aux"]
rt = _hip.Runtime.get()
nf = _tables.frame_count(len(x), fs, args[2])
tp = _tables.frame_times(nf, args[2])
for mode in ("estimate+retry", "hinted", "safe"):
    batch = rt.make_batch([0, len(x)], [0, nf])
    if mode == "hinted":
        batch.flat_samples = [flat_samples(x)]
    xd, tpd = rt.to_device(x), rt.to_device(tp)
    f0, vuv, dbg = harvest_device(rt, batch, xd, tpd, fs, *args, debug=True, event_caps='safe' if mode == "safe" else None)
    fl = rt.take_flags()
    if fl[1]:
        f0, vuv, dbg = harvest_device(rt, batch, xd, tpd, fs, *args, debug=True, event_caps=counted_event_caps(rt))
        assert rt.take_flags() == [0] * 16
    y = dbg["y"].cpu().numpy()[: len(aux["y"])]
    nb = aux["raw"].shape[0]
    raw = dbg["raw"].cpu().numpy()[: nb * aux["raw"].shape[1]].reshape(nb, -1)
    f1 = dbg["f0_1ms"].cpu().numpy()[: len(aux["f0_1ms"])]
    d1 = np.abs(f1 - aux["f0_1ms"])
    live = (raw != 0) != (aux["raw"] != 0)
    print(mode, "overflowed" if fl[1] else "", "| y diff %.3g" % np.max(np.abs(y - aux["y"])), "| raw live mismatch", int(live.sum()), "of", int((aux["raw"] != 0).sum()),
          "| f0_1ms: voiced mismatch", int(np.sum((f1 != 0) != (aux["f0_1ms"] != 0))), "frames > 1e-6:", int((d1 > 1e-6).sum()), "max %.3g" % d1.max(),
          "| final vuv mismatch", int(np.sum(vuv.cpu().numpy() != o["vuv"])), "f0 max diff %.3g" % np.max(np.abs(f0.cpu().numpy() - o["f0"])))
bad = np.nonzero(d1 > 1e-6)[0]
print("1 ms frames that differ:", bad[:40], "of", len(f1), "; utterance spans frames", int(c["shape"]["pad_head"] * 1000), "to", int((c["shape"]["pad_head"] + c["shape"]["seconds"]) * 1000))
for i in bad[:10]:
    print("   frame", i, "ours", f1[i], "oracle", aux["f0_1ms"][i])
if live.any():
    ch, fr = np.nonzero(live)
    print("raw live mismatches: channels", np.unique(ch)[:30], "frames", fr.min(), "..", fr.max(), "; histogram over frames (100 ms bins):", np.histogram(fr, bins=np.arange(0, len(f1) + 100, 100))[0])
for key in ("refined_f0", "pruned_f0", "overlapped"):
    print(key, getattr(aux[key], "shape", None))
