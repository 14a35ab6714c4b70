"""Probe (GPU, subprocess): calls with arguments no analysis can be made of — each must raise a Python exception (or return
something finite), never crash the process or hang.  Prints one line per call and PROBE DONE."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)
from world import _hip
from world._synthetic import synth_utterance
from world.batch import WorldBatch
from world.main import World

fs = 16000
x = synth_utterance(2, fs, 0.3)
w = World()
wb = WorldBatch()
good = w.encode(fs, x, f0_method="dio")

def call(name, fn):
    try:
        r = fn()
        print("%-44s returned %s" % (name, type(r).__name__), flush=True)
    except BaseException as e:  # noqa: BLE001
        print("%-44s raised %s: %s" % (name, type(e).__name__, str(e).replace("\n", " ")[:90]), flush=True)
    try:
        _hip.Runtime.get().take_flags()
        wb.rt.take_flags()
    except Exception:  # noqa: BLE001
        pass

call("empty waveform", lambda: w.encode(fs, np.zeros(0), f0_method="dio"))
call("empty waveform (harvest)", lambda: w.encode(fs, np.zeros(0)))
call("one sample", lambda: w.encode(fs, np.ones(1), f0_method="dio"))
call("31 samples (harvest)", lambda: w.encode(fs, x[:31]))
call("40 samples (harvest)", lambda: w.encode(fs, x[:40]))
call("40 samples (dio)", lambda: w.encode(fs, x[:40], f0_method="dio"))
call("fs = 0", lambda: w.encode(0, x, f0_method="dio"))
call("fs = -16000", lambda: w.encode(-16000, x, f0_method="dio"))
call("fs = 1000", lambda: w.encode(1000, x, f0_method="dio"))
call("fs = 1e9", lambda: w.encode(1000000000, x, f0_method="dio"))
call("frame_period = 0", lambda: w.encode(fs, x, f0_method="dio", frame_period=0))
call("frame_period = -5", lambda: w.encode(fs, x, f0_method="dio", frame_period=-5))
call("f0_floor > f0_ceil (dio)", lambda: w.encode(fs, x, f0_method="dio", f0_floor=500, f0_ceil=100))
call("f0_floor > f0_ceil (harvest)", lambda: w.encode(fs, x, f0_floor=500, f0_ceil=100))
call("f0_floor = 0 (harvest)", lambda: w.encode(fs, x, f0_floor=0))
call("f0_floor = 1 (harvest)", lambda: w.encode(fs, x, f0_floor=1))
call("f0_ceil = fs (harvest)", lambda: w.encode(fs, x, f0_ceil=fs))
call("fft_size = 1000", lambda: w.encode(fs, x, f0_method="dio", fft_size=1000))
call("fft_size = 64", lambda: w.encode(fs, x, f0_method="dio", fft_size=64))
call("fft_size = 65536", lambda: w.encode(fs, x, f0_method="dio", fft_size=65536))
call("unknown f0_method", lambda: w.encode(fs, x, f0_method="yin"))
call("2-D waveform", lambda: w.encode(fs, np.stack([x, x]), f0_method="dio"))
call("integer waveform", lambda: w.encode(fs, (x * 32767).astype(np.int16), f0_method="dio"))
call("channels_in_octave = 0", lambda: w.encode(fs, x, f0_method="dio", channels_in_octave=0))
call("target_fs > fs", lambda: w.encode(fs, x, f0_method="dio", target_fs=32000))
call("encode_batch of nothing", lambda: w.encode_batch(fs, []))
call("encode_batch with an empty utterance", lambda: w.encode_batch(fs, [x, np.zeros(0)], f0_method="dio"))
call("decode of a dict without f0", lambda: w.decode({k: v for k, v in good.items() if k != "f0"}))
call("decode with a transposed spectrogram", lambda: w.decode(dict(good, spectrogram=good["spectrogram"].T.copy())))
call("decode with f0 of another length", lambda: w.decode(dict(good, f0=good["f0"][:-3].copy())))
call("decode with all-zero f0", lambda: w.decode(dict(good, f0=good["f0"] * 0, vuv=good["vuv"] * 0)))
call("decode with negative f0", lambda: w.decode(dict(good, f0=-good["f0"])))
call("decode with f0 = 1e6", lambda: w.decode(dict(good, f0=good["f0"] * 0 + 1e6)))
call("decode with decreasing frame times", lambda: w.decode(dict(good, temporal_positions=good["temporal_positions"][::-1].copy())))
call("decode_batch of mixed rates", lambda: w.decode_batch([dict(good), dict(good, fs=22050)]))
call("scale_pitch by 0", lambda: w.decode(w.scale_pitch(dict(good), 0.0)))
call("scale_duration by 0", lambda: w.decode(w.scale_duration(dict(good), 0.0)))
call("scale_duration by -1", lambda: w.decode(w.scale_duration(dict(good), -1.0)))
call("WorldBatch.encode of ragged with a tiny one", lambda: wb.encode([x, x[:33]], fs, f0_method="harvest"))
print("PROBE DONE")
