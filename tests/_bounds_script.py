"""Helper of tests/test_hip_bounds.py: runs in a process whose WH_LIB is the BOUNDS build of the library
(tools/build_variants.py bounds=...:-DWH_BOUNDS=1; world/_hip.py loads WH_LIB).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel_rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2) / np.mean(b ** 2)))


def main():
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.cheaptrick import cheaptrick

    out = {"bounds_build": _hip.bounds_build()}
    rt = _hip.Runtime.get()
    # ---- positive control: the checker sees a store one past a four-element buffer -----------------------------------
    _hip.check(rt.lib.wh_bounds_selftest(rt.ctx, rt.stream()))
    flags = rt.take_flags()
    out["selftest_flag"] = flags[_hip.FLAG_OOB]
    out["selftest_record"] = list(_hip.bounds_last())
    out["clean_after"] = rt.take_flags()[_hip.FLAG_OOB], list(_hip.bounds_last())
    # ---- the CheapTrick fixtures (reference outputs) -----------------------------------------------------------------
    fix = {}
    for tag in ("syn16k", "syn48k"):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_%s.npz" % tag)))
        fs = int(g["fs"])
        src = {"f0": g["stonemask_f0"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy()}
        res = cheaptrick(g["x"], fs, src)
        fix[tag] = {"rel_rms": rel_rms(res["spectrogram"], g["ct_spectrogram"]),
                    "flag": rt.take_flags()[_hip.FLAG_OOB], "record": list(_hip.bounds_last())}
    out["fixtures"] = fix
    # ---- f0 sweep: constant contours from far below the floor to beyond fs/2, three rates, transforms 512 ... 4096 ----
    sweep = []
    for fs, fft in ((16000, 1024), (16000, 512), (22050, 1024), (48000, 2048), (96000, 4096)):
        x = synth_utterance(5, fs, 0.6)
        n = int(1000 * len(x) / fs / 5 + 1)
        tp = np.arange(n) * 0.005
        for f0 in (1.0, 20.0, 47.0, 70.0, 71.0, 123.4, 250.0, 499.9, 800.0, 1500.0, 0.45 * fs, 0.4999 * fs, 0.5 * fs,
                   0.75 * fs, 1.5 * fs):
            src = {"f0": np.full(n, f0), "vuv": np.ones(n), "temporal_positions": tp.copy()}
            res = cheaptrick(x, fs, src, fft_size=fft)
            fl = rt.take_flags()
            sweep.append({"fs": fs, "fft": fft, "f0": f0, "flag": fl[_hip.FLAG_OOB], "record": list(_hip.bounds_last()),
                          "finite": bool(np.all(np.isfinite(res["spectrogram"])))})
    out["sweep_bad"] = [s for s in sweep if s["flag"] or not s["finite"]]
    out["sweep_cases"] = len(sweep)
    # ---- D4C / D4C-Requiem / love-train (round 6, second half: their LDS blocks, waveform gathers, twiddle and window tables and
    # output rows are checked pointers too): the same constant contours, voiced everywhere, at every transform length
    # 512 ... 8192 the two entry points reach, fused and stand-alone gate
    from world.d4c import d4c
    from world.d4cRequiem import d4cRequiem
    dsweep = []
    for fs, req, fft in ((16000, False, None), (22050, False, None), (48000, False, None), (96000, False, None), (8000, False, None),
                         (16000, True, None), (16000, True, 512), (48000, True, None), (96000, True, 4096)):
        x = synth_utterance(6, fs, 0.4)
        n = int(1000 * len(x) / fs / 5 + 1)
        tp = np.arange(n) * 0.005
        for f0 in (1.0, 20.0, 47.0, 70.0, 71.0, 123.4, 250.0, 499.9, 800.0, 1500.0, 0.45 * fs, 0.4999 * fs, 0.5 * fs,
                   0.75 * fs, 1.5 * fs):
            src = {"f0": np.full(n, f0), "vuv": np.ones(n), "temporal_positions": tp.copy()}
            res = d4cRequiem(x, fs, src, fft_size=fft) if req else d4c(x, fs, src)
            fl = rt.take_flags()
            dsweep.append({"fs": fs, "requiem": req, "fft": fft, "f0": f0, "flag": fl[_hip.FLAG_OOB],
                           "record": list(_hip.bounds_last()), "nan": bool(np.any(np.isnan(res["aperiodicity"])))})
    out["d4c_sweep_bad"] = [s for s in dsweep if s["flag"]]
    out["d4c_sweep_cases"] = len(dsweep)
    # ---- the off-regime signals of the fuzz tests through both pipelines (tone bursts between digital silence, noise, a
    # chirp, clicks, DC, a 0.2 s and a 1e-8 utterance, two tones, silence) -----------------------------------------------
    from _harvest_script import fuzz_inputs
    wbf = WorldBatch()
    for fs_f in (16000, 48000):
        _, xf = fuzz_inputs(fs_f)
        enc = wbf.encode(xf, fs_f, f0_method="dio")
        wbf.decode_device(enc, seed=2)
        enc = wbf.encode(xf, fs_f, f0_method="harvest", is_requiem=True)
        wbf.decode_device(enc)
    out["fuzz_flags"] = wbf.rt.take_flags()
    out["fuzz_record"] = list(_hip.bounds_last())
    # ---- decode sweep (round 6, second half: response_kernel's two chain buffers, padded response, noise block, ring and row,
    # req_filter_kernel's buffers, run accumulator and row, and the shared minimum-phase chain are checked pointers): frame
    # periods 1 / 5 / 10 ms (Requiem hops of 16 ... 480 samples), pulse rates from a quarter to eight times the contour's
    # (windows further apart than a transform; more pulses than the default capacity: the overflow retry), durations
    # halved and tripled, transforms 1024 ... 4096
    wbd = WorldBatch()
    dec_flags = [0] * 16
    dec_cases = 0
    for fs_d, period in ((16000, 1), (16000, 5), (16000, 10), (48000, 5), (48000, 10), (96000, 5)):
        xd = [synth_utterance(9, fs_d, 0.5), synth_utterance(10, fs_d, 0.3)]
        for req in (False, True):
            for pitch, dur in ((1.0, 1.0), (0.25, 1.0), (8.0, 1.0), (1.0, 0.5), (1.5, 3.0)):
                enc = wbd.encode(xd, fs_d, f0_method="dio", frame_period=period, is_requiem=req)
                if pitch != 1.0:
                    enc.scale_pitch(pitch)
                if dur != 1.0:
                    enc.scale_duration(dur)
                wbd.decode_device(enc, seed=4)  # (check=True: a pulse overflow is retried with the safe capacity)
                dec_flags = [a | b for a, b in zip(dec_flags, wbd.rt.take_flags())]
                dec_cases += 1
    out["decode_sweep_flags"] = dec_flags
    out["decode_sweep_record"] = list(_hip.bounds_last())
    out["decode_sweep_cases"] = dec_cases
    # ---- config 2 at full size: 64 x 10 s through every stage of the DIO path + decode, Harvest + Requiem on 8 ----------
    xs = [synth_utterance(u, 16000, 10.0) for u in range(64)]
    wb = WorldBatch()
    enc = wb.encode(xs, 16000, f0_method="dio", check=False)
    y, _ = wb.decode_device(enc, seed=3, check=False)
    out["config2_flags"] = wb.rt.take_flags()
    out["config2_record"] = list(_hip.bounds_last())
    enc = wb.encode(xs[:8], 16000, f0_method="harvest", is_requiem=True, check=False)
    y, _ = wb.decode_device(enc, check=False)
    out["harvest_flags"] = wb.rt.take_flags()
    out["harvest_record"] = list(_hip.bounds_last())
    x48 = [synth_utterance(75, 48000, 3.0)]
    enc = wb.encode(x48, 48000, f0_method="harvest", check=False)
    y, _ = wb.decode_device(enc, seed=1, check=False)
    out["cfg5_flags"] = wb.rt.take_flags()
    out["cfg5_record"] = list(_hip.bounds_last())
    print("BOUNDS_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
