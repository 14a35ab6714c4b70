import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _harvest_script import fuzz_inputs
from oracle import pitch_harvest
from world import _hip, _tables
from world.harvest import harvest_device, counted_event_caps

fs, xs = fuzz_inputs()
rt = _hip.Runtime.get()
which = [int(a) for a in sys.argv[1:]] or range(len(xs))
for u in which:
    x = xs[u]
    if not np.any(x):
        continue
    o = pitch_harvest.harvest_np(x, fs, return_aux=True)
    aux = o["aux"]
    nf = _tables.frame_count(len(x), fs, 5)
    tp = _tables.frame_times(nf, 5)
    batch = rt.make_batch([0, len(x)], [0, nf])
    xd, tpd = rt.to_device(x), rt.to_device(tp)
    f0, vuv, dbg = harvest_device(rt, batch, xd, tpd, fs, debug=True)
    fl = rt.take_flags()
    if fl[1]:
        f0, vuv, dbg = harvest_device(rt, batch, xd, tpd, fs, debug=True, event_caps=counted_event_caps(rt))
        assert rt.take_flags() == [0] * 16
    y = dbg["y"].cpu().numpy()[: len(aux["y"])]
    nb = aux["raw"].shape[0]
    raw = dbg["raw"].cpu().numpy()[: nb * aux["raw"].shape[1]].reshape(nb, -1)
    live = (raw != 0) != (aux["raw"] != 0)
    f1 = dbg["f0_1ms"].cpu().numpy()[: len(aux["f0_1ms"])]
    both = (raw != 0) & (aux["raw"] != 0)
    print("signal", u, "overflowed" if fl[1] else "", "| y max diff %.3g (scale %.3g)" % (np.max(np.abs(y - aux["y"])), np.max(np.abs(aux["y"]))),
          "| raw live mismatches", int(live.sum()), "of", int((aux["raw"] != 0).sum()), "live; max diff where both %.3g" % (np.max(np.abs(raw - aux["raw"])[both]) if both.any() else 0),
          "| f0_1ms voiced mismatch", int(np.sum((f1 != 0) != (aux["f0_1ms"] != 0))), "| vuv mismatch", int(np.sum(vuv.cpu().numpy() != o["vuv"])), flush=True)
    if live.any():
        ch, fr = np.nonzero(live)
        print("   first mismatching (channel, frame):", list(zip(ch[:12].tolist(), fr[:12].tolist())), "channels hit:", np.unique(ch)[:20].tolist(), "frames range", fr.min(), fr.max())
        for c, f in list(zip(ch, fr))[:6]:
            print("     ch %d fr %d ours %.6f oracle %.6f" % (c, f, raw[c, f], aux["raw"][c, f]))
    if both.any():
        d = np.abs(raw - aux["raw"]) * both
        idx = np.argsort(d.ravel())[::-1][:10]
        print("   largest differences where both are live:")
        for i in idx:
            c, f = divmod(int(i), raw.shape[1])
            print("     ch %d fr %d ours %.9f oracle %.9f" % (c, f, raw[c, f], aux["raw"][c, f]))
        print("   count of |diff| > 1e-6 where both live:", int((d > 1e-6).sum()))
