"""Probe (GPU, run under a short timeout): waveforms holding NaN / Inf through both pipelines — the calls must RETURN (whatever the
numbers are; the reference raises or returns NaN on such input), never hang."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)
from world import _hip
from world._synthetic import synth_utterance
from world.batch import WorldBatch

fs = 16000
wb = WorldBatch()
base = synth_utterance(3, fs, 0.4)
for name, edit in (("one NaN", lambda x: x.__setitem__(3000, np.nan)), ("a run of NaN", lambda x: x.__setitem__(slice(2000, 2600), np.nan)),
                   ("+Inf", lambda x: x.__setitem__(1500, np.inf)), ("-Inf and NaN", lambda x: (x.__setitem__(100, -np.inf), x.__setitem__(5000, np.nan))),
                   ("all NaN", lambda x: x.__setitem__(slice(None), np.nan)), ("1e300", lambda x: x.__setitem__(slice(None), x * 1e300))):
    for method, req in (("dio", False), ("harvest", True)):
        x = base.copy()
        edit(x)
        t = time.time()
        try:
            enc = wb.encode([x, base], fs, f0_method=method, is_requiem=req, check=False)
            y, _ = wb.decode_device(enc, seed=1, check=False)
            y.cpu()
            flags = wb.rt.take_flags()
            d = enc.to_dicts()
            print("%-14s %-8s returned in %.2f s; flags %s; clean neighbour finite: %s; voiced frames %d" % (
                name, method, time.time() - t, [i for i, f in enumerate(flags) if f], bool(np.isfinite(d[1]["spectrogram"]).all()), int(np.nansum(d[0]["vuv"]))), flush=True)
        except _hip.WorldHipError as e:
            print("%-14s %-8s raised %s" % (name, method, str(e)[:120]), flush=True)
            wb.rt.take_flags()
print("PROBE DONE")
