"""ORACLE (test infrastructure) — facade mirroring world/main.py:106-214 on top of the NumPy
restatement.  Used by tests, smoke() and bench.py's cpu_baseline leg only."""
import numpy as np

from . import aperiodicity, envelope, pitch_dio, pitch_harvest, resynth


def encode_np(fs, x, f0_method="harvest", f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000,
              frame_period=5, allowed_range=0.1, fft_size=None, is_requiem=False):
    """world/main.py:106-152."""
    if fft_size is not None:
        f0_floor = 3.0 * fs / fft_size
    if f0_method == "dio":
        src = pitch_dio.dio_np(x, fs, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, allowed_range)
        src["f0"] = pitch_dio.stonemask_np(x, fs, src["temporal_positions"], src["f0"])
    elif f0_method == "harvest":
        src = pitch_harvest.harvest_np(x, fs, f0_floor, f0_ceil, frame_period)
    elif f0_method == "swipe":  # world/main.py:134-135: default dt = 5 ms whatever frame_period is
        from . import pitch_swipe

        src = pitch_swipe.swipe_np(fs, x, [f0_floor, f0_ceil], sTHR=0.3)
    else:
        raise Exception
    tp, vuv = src["temporal_positions"], src["vuv"]
    spec, ps, f0_ct = envelope.cheaptrick_np(x, fs, src["f0"], vuv, tp, fft_size=fft_size)
    if is_requiem:
        ap, f0_out = aperiodicity.d4c_requiem_np(x, fs, f0_ct, vuv, tp, fft_size=fft_size)
    else:
        ap, _, f0_out = aperiodicity.d4c_np(x, fs, f0_ct, vuv, tp, fft_size_for_spectrum=fft_size)
    return {"temporal_positions": tp, "vuv": vuv, "fs": fs, "f0": f0_out, "aperiodicity": ap,
            "ps spectrogram": ps, "spectrogram": spec, "is_requiem": is_requiem}


def decode_np(dat, noise=None, seeds=None, cursor=None):
    """world/main.py:198-214."""
    if dat["is_requiem"]:
        if seeds is None:
            seeds = resynth.seeds_np(dat["fs"])
        y, _ = resynth.synthesis_requiem_np(dat["f0"], dat["vuv"], dat["temporal_positions"], dat["spectrogram"],
                                            dat["aperiodicity"], dat["fs"], seeds, cursor)
    else:
        y = resynth.synthesis_np(dat["f0"], dat["vuv"], dat["temporal_positions"], dat["spectrogram"],
                                 dat["aperiodicity"], dat["fs"], noise=noise)
    m = np.max(np.abs(y))
    if m > 1.0:
        y = y / m
    dat["out"] = y
    return dat
