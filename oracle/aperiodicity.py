"""ORACLE (test infrastructure) — D4C ("Love Train") and D4C-Requiem band aperiodicity.

Restates world/d4c.py:10-233 and world/d4cRequiem.py:9-215 (their helpers are duplicates) with
frames processed row-wise.  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np

from . import common as C


def _pow2_at_least(v: float) -> int:
    return int(2 ** np.ceil(np.log2(v)))


def _window_frames(x, fs, f0, pos, half_length, blackman):
    """world/d4c.py:92-110 — note the sub-sample phase term in the window argument."""
    seg, rel, valid, hwl = C.gather_frames(x, fs, f0, pos, half_length)
    t = rel / fs / half_length + ((pos * fs - np.trunc(pos * fs + 0.5)) / fs)[:, None]
    arg = np.pi * t * f0[:, None]
    if blackman:
        w = 0.08 * np.cos(arg * 2) + 0.5 * np.cos(arg) + 0.42
    else:
        w = 0.5 * np.cos(arg) + 0.5
    w = np.where(valid, w, 0.0)
    return C.remove_dc(seg, w, valid, hwl), valid, hwl


def love_train(x, fs, f0, pos, threshold):
    """VUV gate — world/d4c.py:68-88.  f0 == 0 → 0 without touching the signal."""
    f0 = np.asarray(f0, dtype=np.float64)
    out = np.zeros(len(f0), dtype=np.int64)
    act = np.nonzero(f0 != 0)[0]
    if len(act) == 0:
        return out
    lowest = 40.0
    nfft = _pow2_at_least(3 * fs / lowest + 1)
    b0 = int(np.ceil(100 / (fs / nfft)) + 1)
    b1 = int(np.ceil(4000 / (fs / nfft)) + 1)
    b2 = int(np.ceil(7900 / (fs / nfft)) + 1)
    wave, _, _ = _window_frames(x, fs, np.maximum(f0[act], lowest), np.asarray(pos)[act], 1.5, True)
    p = np.abs(np.fft.fft(wave, nfft, axis=1)) ** 2
    p[:, :b0] = 0.0
    cum = np.cumsum(p, axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        out[act] = (cum[:, b1 - 1] / cum[:, b2 - 1]) > threshold
    return out


def _centroid(wave, nfft):
    """world/d4c.py:146-153: -Im(W)Re(S)+Im(S)Re(W) with W = FFT(-j n x), n 1-based."""
    n = np.arange(1, wave.shape[1] + 1)[None, :]
    xn = wave / np.sqrt(np.sum(wave ** 2, axis=1))[:, None]
    s = np.fft.fft(xn, nfft, axis=1)
    w = np.fft.fft(-xn * n * 1j, nfft, axis=1)
    return -w.imag * s.real + s.imag * w.real


def coarse_aperiodicity(x, fs, f0, pos, nfft, interval, n_bands, window):
    """estimate_one_slice for the rows of f0 (all non-zero) — world/d4c.py:114-209.
    Returns (F, n_bands) positive dB values."""
    f0 = np.asarray(f0, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    # static centroid from two frames at ±T0/4 (d4c.py:132-142)
    w1, _, _ = _window_frames(x, fs, f0, pos + 1 / f0 / 4, 2, True)
    w2, _, _ = _window_frames(x, fs, f0, pos - 1 / f0 / 4, 2, True)
    centroid = C.low_band_replica(_centroid(w1, nfft) + _centroid(w2, nfft), fs, nfft, f0, 1.2 * f0)
    # smoothed power (d4c.py:157-161)
    wh, _, _ = _window_frames(x, fs, f0, pos, 2, False)
    power = C.low_band_replica(np.abs(np.fft.fft(wh, nfft, axis=1)) ** 2, fs, nfft, f0, 1.2 * f0)
    smoothed = C.mirror_half(C.cumsum_band_mean(power, fs, nfft, f0) / f0[:, None])
    # group-delay shaping (d4c.py:165-174)
    with np.errstate(invalid="ignore", divide="ignore"):
        gd = centroid / smoothed
    gd = C.mirror_half(C.cumsum_band_mean(gd, fs, nfft, f0 / 2) / (f0 / 2)[:, None])
    gd_b = C.cumsum_band_mean(gd, fs, nfft, f0) / f0[:, None]
    gd = C.mirror_half(gd[:, : nfft // 2 + 1] - gd_b)
    # band-wise ratio (d4c.py:192-209)
    boundary = int(nfft / len(window) * 8 + 0.5)
    half = int(np.floor(len(window) / 2))
    out = np.zeros((len(f0), n_bands))
    for b in range(n_bands):
        centre = int(np.floor(interval * (b + 1) / (fs / nfft)))
        seg = gd[:, centre - half : centre + half + 1] * window[None, :]
        p = np.abs(np.fft.fft(seg, nfft, axis=1)) ** 2
        cum = np.cumsum(np.sort(p[:, : nfft // 2 + 1], axis=1), axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[:, b] = -10 * np.log10(cum[:, nfft // 2 - boundary - 1] / cum[:, -1])
    return out


def d4c_np(x, fs, f0, vuv, temporal_positions, threshold=0.85, fft_size_for_spectrum=None):
    """world/d4c.py:10-64.  Returns (aperiodicity (K,F), coarse_ap (nap,F), f0_out (F,))."""
    x = np.asarray(x, dtype=np.float64)
    low = 47.0
    nfft = _pow2_at_least(4 * fs / low + 1)
    if fft_size_for_spectrum is None:
        fft_size_for_spectrum = _pow2_at_least(3 * fs / 71 + 1)
    interval = 3000
    if fs < 16000:
        interval = 2000
    f0o = np.array(f0, dtype=np.float64, copy=True)
    f0o[np.asarray(vuv) == 0] = 0
    pos = np.asarray(temporal_positions, dtype=np.float64)
    nap = int(np.floor(np.min([15000, fs / 2 - interval]) / interval))
    assert nap > 0
    window = C.nuttall_window(np.floor(interval / (fs / nfft)) * 2 + 1)
    k = fft_size_for_spectrum // 2 + 1
    ap = np.full((k, len(f0o)), 1 - 0.000000000001)
    coarse_dbg = np.zeros((nap, len(f0o)))
    gate = love_train(x, fs, f0o, pos, threshold)
    act = np.nonzero(gate)[0]
    if len(act):
        cur = np.maximum(low, f0o[act])
        ca = coarse_aperiodicity(x, fs, cur, pos[act], nfft, interval, nap, window)
        ca = np.maximum(0, ca - ((cur - 100) * 2 / 100)[:, None])
        coarse_dbg[:, act] = -ca.T
        freq_axis = np.arange(fft_size_for_spectrum / 2 + 1) * fs / fft_size_for_spectrum
        coarse_axis = np.r_[np.arange(nap + 1) * interval, fs / 2]
        nodes = np.concatenate([np.full((len(act), 1), -60.0), -ca, np.full((len(act), 1), -0.000000000001)], axis=1)
        ap[:, act] = 10 ** (C.lerp_extrap(coarse_axis, nodes.T, freq_axis) / 20)
    return ap, coarse_dbg, f0o


def d4c_requiem_np(x, fs, f0, vuv, temporal_positions, threshold=0.85, fft_size=None):
    """world/d4cRequiem.py:9-44.  Returns (band_aperiodicity dB (nap+2,F), f0_out)."""
    x = np.asarray(x, dtype=np.float64)
    low = 47.0
    if fft_size is None:
        fft_size = _pow2_at_least(3 * fs / low + 1)
    nfft = int(fft_size)
    interval = 3000
    f0o = np.array(f0, dtype=np.float64, copy=True)
    f0o[np.asarray(vuv) == 0] = 0
    pos = np.asarray(temporal_positions, dtype=np.float64)
    nap = int(np.floor(np.min([15000, fs / 2 - interval]) / interval))
    assert nap > 0
    window = C.nuttall_window(np.floor(interval / (fs / nfft)) * 2 + 1)
    band = np.zeros((nap + 2, len(f0o)))
    band[0, :] = -60
    band[-1, :] = -0.000000000001
    gate = love_train(x, fs, f0o, pos, threshold)
    band[:, gate == 0] = -0.000000000001
    act = np.nonzero(gate)[0]
    if len(act):
        cur = np.maximum(low, f0o[act])
        ca = coarse_aperiodicity(x, fs, cur, pos[act], nfft, interval, nap, window)
        band[1:-1, act] = -np.maximum(0, ca - ((cur - 100) * 2 / 100)[:, None]).T
    return band, f0o
