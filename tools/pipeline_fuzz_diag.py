"""Diagnostic (GPU): the off-regime signals of tests/_harvest_script.py through both whole pipelines against the oracle."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _harvest_script import fuzz_inputs
from oracle import api as oapi
from world.batch import WorldBatch

def rel_rms(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-300))

fs, xs = fuzz_inputs()
wb = WorldBatch()
for method, req in (("dio", False), ("harvest", True)):
    print("=====", method, "requiem" if req else "")
    try:
        enc = wb.encode(xs, fs, f0_method=method, is_requiem=req)
        dicts = enc.to_dicts()
    except Exception as e:
        print("batch encode raised:", repr(e)); traceback.print_exc(); continue
    rng = np.random.RandomState(5)
    noise = [rng.randn(2 * len(x) + 4096) for x in xs]
    for u, x in enumerate(xs):
        d = dicts[u]
        try:
            o = oapi.encode_np(fs, x, f0_method=method, is_requiem=req)
        except Exception as e:
            print(u, "oracle encode raised", repr(e), "| ours voiced", int(d['vuv'].sum())); continue
        line = "%d vuv mismatch %d (voiced %d) f0 %.2g spec %.2g ap %.2g" % (
            u, int(np.sum(d['vuv'] != o['vuv'])), int(o['vuv'].sum()), rel_rms(d['f0'], o['f0']),
            rel_rms(d['spectrogram'], o['spectrogram']), rel_rms(d['aperiodicity'], o['aperiodicity']))
        fin = np.isfinite(d['spectrogram']).all() and np.isfinite(d['aperiodicity']).all()
        print(line, "finite" if fin else "NOT FINITE", "| oracle finite", bool(np.isfinite(o['spectrogram']).all() and np.isfinite(o['aperiodicity']).all()), flush=True)
    # decode, one utterance at a time (an utterance without a pulse makes the reference assert)
    for u, x in enumerate(xs):
        try:
            e1 = wb.encode([x], fs, f0_method=method, is_requiem=req)
            d1 = e1.to_dicts()[0]
            if req:
                from world.get_seeds_signals import get_seeds_signals
                import random
                random.seed(1); np.random.seed(1)
                seeds = get_seeds_signals(fs)
                y, y_off = wb.decode_device(e1, seeds=seeds)
                yo = oapi.decode_np(dict(d1), seeds=seeds)["out"]
            else:
                y, y_off = wb.decode_device(e1, noise=[noise[u]])
                yo = oapi.decode_np(dict(d1), noise=noise[u])['out']
            y = y.cpu().numpy()
            print(u, "decode len", len(y), len(yo), "rel rms %.2g" % rel_rms(y, yo), "peak %.3g" % np.max(np.abs(yo)), flush=True)
        except Exception as e:
            print(u, "decode raised", type(e).__name__, str(e)[:200], flush=True)
            try:
                wb.rt.take_flags()
            except Exception:
                pass
