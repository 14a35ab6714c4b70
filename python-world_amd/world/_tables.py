"""Host-built constant tables handed to the HIP library as data.

These are the places where the reference's result depends on Python/NumPy *host* semantics
(string-formatted rounding, an argmax tie decided by the last bit of np.cos), so they are
evaluated here with the same NumPy expressions and uploaded.
"""
import math

import functools

import numpy as np
from scipy.signal.windows import hann


def nuttall(n):
    """4-term Nuttall window through the (1x4)@(4xN) product the reference uses (world/dio.py:208-212),
    so that the two mathematically equal maxima of an even-length window compare the same way."""
    t = np.asmatrix(np.arange(n) * 2 * math.pi / (n - 1))
    coefs = np.array([0.355768, -0.487396, 0.144232, -0.012604])
    w = coefs @ np.cos(np.matrix([0, 1, 2, 3]).T @ t)
    return np.squeeze(np.asarray(w))


def dio_tables(f0_floor, f0_ceil, channels_in_octave, target_fs):
    """Band list, Nuttall low-pass taps, delay indices and the low-cut FIR of DIO
    (world/dio.py:32-34,80-83,129-131)."""
    bands = np.arange(math.ceil(np.log2(f0_ceil / f0_floor) * channels_in_octave)) + 1
    bands = f0_floor * (2.0 ** (bands / channels_in_octave))
    taps, lens, bias = [], [], []
    for bf in bands:
        half = int(target_fs / bf / 2 + 0.5)
        w = nuttall(half * 4)
        taps.append(w)
        lens.append(len(w))
        bias.append(int(w.argmax()))
    cut = int(target_fs / 50 + 0.5)
    h = hann(2 * cut + 3)[1:-1]
    h = -h / np.sum(h)
    h[cut] += 1
    return {
        "band_f0": np.ascontiguousarray(bands, dtype=np.float64),
        "band_len": np.asarray(lens, dtype=np.int32),
        "band_bias": np.asarray(bias, dtype=np.int32),
        "band_taps": np.ascontiguousarray(np.concatenate(taps), dtype=np.float64),
        "lowcut": np.ascontiguousarray(h, dtype=np.float64),
        "lowcut_half": cut,
    }


_QT_CACHE = {}


def quantised_times(fs, kmax):
    """table[k+kmax] = float('%.4f' % (k/fs)) for k in [-kmax, kmax] (world/stonemask.py:38)."""
    key = (float(fs), int(kmax))
    t = _QT_CACHE.get(key)
    if t is None:
        t = np.array([float("{0:.4f}".format(e)) for e in (np.arange(-kmax, kmax + 1) / fs)], dtype=np.float64)
        if len(_QT_CACHE) >= 64:  # (bounded: a long-running caller with ever new rates / bounds starts over)
            _QT_CACHE.clear()
        _QT_CACHE[key] = t
    return t


def frame_count(n_samples, fs, frame_period):
    return int(1000 * n_samples / fs / frame_period + 1)


def frame_times(n_frames, frame_period):
    return np.arange(0, n_frames) * frame_period / 1000


_HV_CACHE = {}


def harvest_tables(fs, f0_floor, f0_ceil):
    """Decimation filter, channel list and band-pass FIRs of Harvest (world/harvest.py:19-29,59,253-256,599)."""
    from decimal import ROUND_HALF_UP, Decimal

    from scipy import signal

    key = (float(fs), float(f0_floor), float(f0_ceil))
    t = _HV_CACHE.get(key)
    if t is not None:
        return t
    target_fs = 8000
    r = int(fs / target_fs + 0.5)
    if fs <= target_fs:
        r = 1
    fs_d = fs / r if r > 1 else fs
    # the filter runs whenever fs > target_fs — also at a ratio of 1 (8 kHz < fs < 12 kHz: harvest.py:60 branches on the
    # rate, and decimate_matlab(x, 1) still low-pass filters at 0.8 of Nyquist); zeros: no filter (wh_harvest reads a0)
    if fs > target_fs:
        b, a = signal.cheby1(3, 0.05, 0.8 / r)
        ba = np.concatenate([b, a]).astype(np.float64)
        zi = np.asarray(signal.lfilter_zi(b, a), dtype=np.float64)
    else:
        ba = np.zeros(8)
        zi = np.zeros(3)
    lo = f0_floor * 0.9
    hi = f0_ceil * 1.1
    bands = np.arange(np.ceil(np.log2(hi / lo) * 40)) + 1
    bands = (2.0 ** (bands / 40)) * lo
    halves, taps = [], []
    for bf in bands:
        h = int(Decimal(fs_d / bf * 2).quantize(0, ROUND_HALF_UP))
        halves.append(h)
        taps.append(nuttall(h * 2 + 1) * np.cos(2 * math.pi * bf * np.arange(-h, h + 1) / fs_d))
    t = {"r": r, "fs_d": fs_d, "ba": np.ascontiguousarray(ba), "zi": np.ascontiguousarray(zi),
         "band_f0": np.ascontiguousarray(bands, dtype=np.float64), "band_half": np.asarray(halves, dtype=np.int32),
         "band_taps": np.ascontiguousarray(np.concatenate(taps), dtype=np.float64)}
    if len(_HV_CACHE) >= 32:  # (bounded, ~1 MB an entry: a caller that varies the search range without end starts over)
        _HV_CACHE.clear()
    _HV_CACHE[key] = t
    return t


@functools.lru_cache(maxsize=16)
def warp_tables(k_bins, factor):
    """NumPy's search for World.warp_spectrum (world/main.py:191-196), done once per (K, factor): for query points
    x_k = (k/K)**factor on the knots xp_k = k/K, the interval index j with xp[j] <= x < xp[j+1], x - xp[j] and
    xp[j+1] - xp[j]; a zero denominator marks the cases in which np.interp returns fp[j] itself (exact knot, last
    knot, or x beyond it)."""
    xp = np.arange(0, k_bins) / k_bins
    x = xp ** factor
    j = np.searchsorted(xp, x, side="right") - 1
    j = np.clip(j, 0, k_bins - 1)
    copy = (j >= k_bins - 1) | (xp[j] == x) | (x > xp[-1])
    jn = np.minimum(j + 1, k_bins - 1)
    den = np.where(copy, 0.0, xp[jn] - xp[j])
    dx = np.where(copy, 0.0, x - xp[j])
    return (np.ascontiguousarray(j, dtype=np.int32), np.ascontiguousarray(dx, dtype=np.float64),
            np.ascontiguousarray(den, dtype=np.float64))
