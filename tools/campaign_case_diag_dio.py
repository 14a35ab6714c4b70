"""Diagnostic (GPU): one case of tools/differential_campaign.py through DIO and StoneMask against the oracle.
    python tools/campaign_case_diag_dio.py <seed> <case index>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import differential_campaign as dc
from oracle import pitch_dio
from world.dio import dio
from world.stonemask import stonemask

seed, idx = int(sys.argv[1]), int(sys.argv[2])
c = dc.draw_case(idx, seed)
print(c)
x = dc.make_input(c)
fs, kw = c["fs"], c["kw"]
a = (kw.get("f0_floor", 71), kw.get("f0_ceil", 800), kw.get("channels_in_octave", 2), 4000, kw.get("frame_period", 5), kw.get("allowed_range", 0.1))
d = dio(x, fs, *a)
od = pitch_dio.dio_np(x, fs, *a)
print("dio: vuv equal", np.array_equal(d["vuv"], od["vuv"]), "f0 max abs diff", float(np.max(np.abs(d["f0"] - od["f0"]))))
sm = stonemask(x, fs, od["temporal_positions"], od["f0"].copy())
osm = pitch_dio.stonemask_np(x, fs, od["temporal_positions"], od["f0"].copy())
dd = np.abs(sm - osm)
bad = np.nonzero(dd > 1e-9 * np.maximum(osm, 1))[0]
print("stonemask on the oracle's DIO contour: frames differing > 1e-9 rel:", bad, "of", len(sm))
for i in bad[:12]:
    print("   frame", i, "t=%.3f" % od["temporal_positions"][i], "dio f0", od["f0"][i], "ours", sm[i], "oracle", osm[i], "rel %.2e" % (dd[i] / osm[i]))
print("utterance spans", c["shape"]["pad_head"], "..", c["shape"]["pad_head"] + c["shape"]["seconds"])
df = np.abs(d["f0"] - od["f0"])
badf = np.nonzero(df > 1e-9)[0]
print("DIO frames differing:", badf)
for i in badf[:12]:
    print("   frame", i, "t=%.3f" % od["temporal_positions"][i], "ours", d["f0"][i], "oracle", od["f0"][i])
for key in ("raw_f0_candidates", "f0_candidates"):
    if key in d and key in od:
        a_, b_ = np.asarray(d[key]), np.asarray(od[key])
        if a_.shape == b_.shape:
            dm = np.abs(a_ - b_)
            w = np.argwhere(dm > 1e-6)
            print(key, a_.shape, "entries differing > 1e-6:", len(w), w[:10].tolist())
            for (bb, ff) in w[:6]:
                print("     band", bb, "frame", ff, "ours", a_[bb, ff], "oracle", b_[bb, ff])
        else:
            print(key, "shapes", a_.shape, b_.shape)
print(sorted(d.keys()), sorted(od.keys()))
