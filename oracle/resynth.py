"""ORACLE (test infrastructure) — pulse-by-pulse overlap-add synthesis and the Requiem variant.

Restates world/synthesis.py:21-250, world/synthesisRequiem.py:12-141 and
world/get_seeds_signals.py:8-73 with the per-pulse / per-frame spectral work batched row-wise.
Only tests/, smoke() and bench.py's cpu_baseline may import this.

Randomness: the reference draws np.random.randn(max(3, noise_size)) once per pulse,
in pulse order; consecutive legacy randn calls are one stream, so drawing the total once and
splitting gives bit-identical noise.  Callers may also hand the noise in (``noise=``) so that
the HIP path and this oracle consume the same samples.
"""
import random
from decimal import ROUND_HALF_UP, Decimal

import numpy as np
from scipy.fft import fft as sp_fft
from scipy.fft import ifft as sp_ifft
from scipy.signal.windows import hann

from . import common as C


def output_length(temporal_positions, fs) -> int:
    """len(np.arange(tp[0], tp[-1]+1/fs, 1/fs)) — world/synthesis.py:39."""
    tp = temporal_positions
    return len(np.arange(tp[0], tp[-1] + 1 / fs, 1 / fs))


def pulse_train(temporal_positions, f0, fs, vuv, default_f0=500):
    """world/synthesis.py:120-140 → (pulse times, 1-based sample indices, fractional shift [s],
    per-sample boolean vuv, time axis)."""
    tp = np.asarray(temporal_positions, dtype=np.float64)
    t = np.arange(tp[0], tp[-1] + 1 / fs, 1 / fs)
    f0_i = C.lerp_extrap(tp, np.asarray(f0, dtype=np.float64), t)
    vuv_i = C.lerp_extrap(tp, np.asarray(vuv, dtype=np.float64), t) > 0.5
    f0_i = f0_i * vuv_i
    f0_i[f0_i == 0] = f0_i[f0_i == 0] + default_f0
    total = np.cumsum(2 * np.pi * f0_i / fs)
    wrap = np.remainder(total, 2 * np.pi)
    at = np.nonzero(np.abs(np.diff(wrap)) > np.pi)[0]
    times = t[:-1][at]
    assert len(times) > 0
    idx = np.array([int(Decimal(e * fs).quantize(0, ROUND_HALF_UP)) for e in times], dtype=np.int64) + 1
    y1 = wrap[idx - 1] - 2.0 * np.pi
    y2 = wrap[idx]
    shift = (-y1 / (y2 - y1)) / fs
    return times, idx, shift, vuv_i, t


def min_phase_spectrum(half_amp: np.ndarray, nfft: int) -> np.ndarray:
    """Rows of K=nfft/2+1 amplitude-like values → rows of nfft complex minimum-phase spectra.
    world/synthesis.py:103-111: cepstrum of log|S|/2, fold (×2 on the upper half, bin 0 kept,
    lower half zeroed), exp(IFFT(.))."""
    full = C.mirror_half(half_amp)
    ceps = np.fft.fft(np.log(np.abs(full)) / 2, axis=1).real
    folded = np.zeros_like(ceps)
    folded[:, nfft // 2 :] = ceps[:, nfft // 2 :] * 2
    folded[:, 0] = ceps[:, 0]
    return np.exp(np.fft.ifft(folded, axis=1))


def _ola(y, start_index_1b, base_index, values):
    """y[clip(idx)-1] += v with NumPy's buffered fancy-index semantics."""
    tgt = np.maximum(1, np.minimum(len(y), start_index_1b + base_index)).astype(np.int64) - 1
    y[tgt] += values


def synthesis_np(f0, vuv, temporal_positions, spectrogram, aperiodicity, fs, noise=None, return_aux=False):
    """world/synthesis.py:21-82.  spectrogram/aperiodicity are (K,F).  ``noise`` — optional 1-D
    stream consumed max(3,noise_size) per pulse in pulse order; default np.random.randn."""
    tp = np.asarray(temporal_positions, dtype=np.float64)
    spectrogram = np.asarray(spectrogram, dtype=np.float64)
    times, idx, shift, vuv_i, t_axis = pulse_train(tp, f0, fs, vuv)
    y = np.zeros(len(t_axis))
    nfft = (spectrogram.shape[0] - 1) * 2
    k = nfft // 2 + 1
    base_index = np.arange(-nfft // 2 + 1, nfft // 2 + 1)
    pos_idx = C.lerp_extrap(tp, np.arange(1, len(tp) + 1, dtype=np.float64), times)
    pos_idx = np.maximum(1, np.minimum(len(tp), pos_idx))
    amp_ap = np.asarray(aperiodicity, dtype=np.float64) ** 2
    amp_p = np.maximum(0.001, 1 - amp_ap)
    dc_base = hann(nfft + 2)[1:-1]
    dc_base = dc_base / np.sum(dc_base)
    coeff = 2.0 * np.pi * fs / nfft

    # spectral parameters per pulse (synthesis.py:144-180)
    lo = np.floor(pos_idx).astype(np.int64) - 1
    hi = np.ceil(pos_idx).astype(np.int64) - 1
    t1 = tp[lo]
    t2 = tp[hi]
    xq = np.maximum(t1, np.minimum(t2, times))
    same = t1 == t2
    with np.errstate(invalid="ignore", divide="ignore"):
        b = np.where(same, 0.0, (xq - t1) / (t2 - t1))
    a = 1 - b

    def blend(m):
        lo_c = m[:, lo].T
        out = a[:, None] * lo_c + b[:, None] * m[:, hi].T
        out[same] = lo_c[same]
        return out

    spec = blend(spectrogram)
    per = blend(amp_p)
    aper = blend(amp_ap)

    nxt = idx[np.minimum(len(idx) - 1, np.arange(len(idx)) + 1)]
    noise_size = nxt - idx
    voiced = (vuv_i[idx - 1] >= 0.5) & (aper[:, 0] <= 0.999)

    # periodic responses (synthesis.py:100-116)
    ps = spec * per
    ps[ps == 0] = C.EPS
    mp = min_phase_spectrum(ps, nfft)[:, :k]
    mp = mp * np.exp(-1j * coeff * shift[:, None] * np.arange(k)[None, :])
    full = np.concatenate([mp, mp[:, -2:0:-1].conj()], axis=1)
    resp_p = np.fft.fftshift(np.fft.ifft(full, axis=1).real, axes=1)
    resp_p = resp_p + dc_base[None, :] * -np.sum(resp_p, axis=1)[:, None]
    resp_p = resp_p * np.sqrt(np.maximum(1, noise_size))[:, None]

    # aperiodic responses (synthesis.py:86-96)
    asp = np.where(voiced[:, None], spec * aper, spec)
    asp[asp == 0] = C.EPS
    mpa = min_phase_spectrum(asp, nfft)
    resp_a = np.fft.fftshift(np.fft.ifft(mpa, axis=1).real, axes=1)

    draws = np.maximum(3, noise_size)
    if noise is None:
        noise = np.random.randn(int(draws.sum()))
    noise = np.asarray(noise, dtype=np.float64)
    assert len(noise) >= draws.sum(), "noise stream too short"
    off = np.concatenate([[0], np.cumsum(draws)])
    for i in range(len(idx)):
        if voiced[i]:
            _ola(y, idx[i], base_index, resp_p[i])
        nz = noise[off[i] : off[i + 1]]
        nz = nz - np.mean(nz)
        _ola(y, idx[i], base_index, np.convolve(nz, resp_a[i])[:nfft])
    if return_aux:
        return y, {"pulse_index": idx, "pulse_shift": shift, "pulse_times": times, "noise_used": int(draws.sum()),
                   "voiced": voiced, "resp_p": resp_p, "resp_a": resp_a}
    return y


# ----------------------------------------------------------------------------------------------
# Requiem: seeds, excitation, frame-wise minimum-phase filtering
# ----------------------------------------------------------------------------------------------

def _short_velvet(n: int) -> np.ndarray:
    """world/get_seeds_signals.py:56-73 (uses the global `random` and np.random streams)."""
    out = np.zeros(n)
    td = 4
    r = int(n // td + 0.5)
    pool = np.ones(r)
    pool[int(r // 2) :] *= -1
    pool *= 2
    for i in range(r):
        j = random.randint(0, r - 1)
        pool[j], pool[i] = pool[i], pool[j]
    out[td * np.arange(r) + np.random.randint(td, size=r)] = pool
    return out


def _velvet(n: int, fs: float) -> np.ndarray:
    """world/get_seeds_signals.py:40-53."""
    short = 8 * C.half_up(np.array([8, 30, 60]) * fs / 48000)  # not truncated: SURVEY Q1(iii)
    buf = np.zeros(n + int(np.max(short)) + 1)
    at = 0
    while True:
        pick = random.randint(0, len(short) - 1)
        ln = int(short[pick])
        buf[at : at + ln] = _short_velvet(ln)
        at += ln
        if at >= n - 1:
            break
    return buf[:n]


def seeds_np(fs, fft_size=None, noise_length=None):
    """world/get_seeds_signals.py:8-38 → {'pulse': (fft, nb), 'noise': (noise_len, nb)}."""
    if fft_size is None:
        fft_size = int(1024 * (2 ** np.ceil(np.log2(fs / 48000))))
    if noise_length is None:
        noise_length = int(2 ** np.ceil(np.log2(fs / 2)))
    w = np.arange(fft_size // 2 + 1) * fs / fft_size
    step = 3000
    nb = int(2 + np.floor(min(15000, fs / 2 - step) / step))
    pulse = np.zeros((fft_size, nb))
    noise = np.zeros((noise_length, nb))
    spec_n = sp_fft(_velvet(noise_length, fs), noise_length)
    for b in range(nb):
        s = 0.5 + 0.5 * np.cos(((w - step * b) / (step * 2)) * 2 * np.pi)
        s[w > step * (b + 1)] = 0
        s[w < step * (b - 1)] = 0
        if b == nb - 1:
            s[w > step * b] = 1
        pulse[:, b] = np.fft.fftshift(sp_ifft(np.r_[s, s[-2:0:-1]]).real)
        noise[:, b] = sp_ifft(spec_n * sp_fft(pulse[:, b], noise_length)).real
    h = hann(fft_size + 2)[1:-1]
    pulse[:, 0] = pulse[:, 0] - np.mean(pulse[:, 0]) * h / np.mean(h)
    return {"pulse": pulse, "noise": noise}


def requiem_pulse_index(temporal_positions, f0, fs, vuv):
    """world/synthesisRequiem.py:104-118."""
    _, idx, _, vuv_i, t = pulse_train_noassert(temporal_positions, f0, fs, vuv)
    return idx, vuv_i, t


def pulse_train_noassert(temporal_positions, f0, fs, vuv):
    tp = np.asarray(temporal_positions, dtype=np.float64)
    t = np.arange(tp[0], tp[-1] + 1 / fs, 1 / fs)
    f0_i = C.lerp_extrap(tp, np.asarray(f0, dtype=np.float64), t)
    vuv_i = C.lerp_extrap(tp, np.asarray(vuv, dtype=np.float64), t) > 0.5
    f0_i = f0_i * vuv_i
    f0_i[f0_i == 0] = f0_i[f0_i == 0] + 500
    wrap = np.remainder(np.cumsum(2 * np.pi * f0_i / fs), 2 * np.pi)
    at = np.nonzero(np.abs(np.diff(wrap)) > np.pi)[0]
    times = t[:-1][at]
    idx = np.array([int(Decimal(e * fs).quantize(0, ROUND_HALF_UP)) for e in times], dtype=np.int64) + 1
    return times, idx, None, vuv_i, t


def requiem_excitation(temporal_positions, fs, f0, vuv, pulse_seed, noise_seed, band_ap_db, cursor=None):
    """world/synthesisRequiem.py:27-71,120-141.  ``cursor``: per-band read position carried between
    calls (the reference keeps it in a function attribute; None = fresh).  Returns (excitation, cursor)."""
    tp = np.asarray(temporal_positions, dtype=np.float64)
    nfft = pulse_seed.shape[0]
    nb = pulse_seed.shape[1]
    base_index = np.arange(-nfft // 2 + 1, nfft // 2 + 1)
    idx, vuv_i, t = requiem_pulse_index(tp, f0, fs, vuv)
    n = len(t)
    ap = np.stack([C.lerp_extrap(tp, 10 ** (np.asarray(band_ap_db[b], dtype=np.float64) / 10), t) for b in range(nb)])
    if cursor is None:
        cursor = np.zeros(noise_seed.shape[1])
    cursor = np.array(cursor, dtype=np.float64, copy=True)
    aper = np.zeros(n)
    nlen = noise_seed.shape[0]
    for b in range(nb):
        rd = np.remainder(np.arange(cursor[b], cursor[b] + n), nlen).astype(np.int64)
        aper += noise_seed[rd, b] * ap[b, :n]
        cursor[b] = rd[-1]
    per = np.zeros(n)
    for i in range(len(idx)):
        p = idx[i] - 1
        if (vuv_i[p] <= 0.5) or (ap[0, p] > 0.999):
            continue
        ns = idx[min(len(idx) - 1, i + 1)] - idx[i]
        resp = _seed_mix(pulse_seed, ap[:, p])
        _ola(per, idx[i], base_index, resp * np.sqrt(max(1, ns)))
    return per + aper, cursor


def _seed_mix(pulse_seed, ap_col):
    """world/synthesisRequiem.py:66-71 (band-by-band accumulation order kept)."""
    out = np.zeros(pulse_seed.shape[0])
    for b in range(pulse_seed.shape[1]):
        out += pulse_seed[:, b] * (1 - ap_col[b])
    return out


def requiem_filter(excitation, spectrogram, temporal_positions, n_frames, fs):
    """world/synthesisRequiem.py:74-101: Hann-windowed excitation frames through the frame's
    minimum-phase spectrum, overlap-added."""
    tp = np.asarray(temporal_positions, dtype=np.float64)
    spectrogram = np.asarray(spectrogram, dtype=np.float64)
    y = np.zeros(len(excitation))
    nfft = (spectrogram.shape[0] - 1) * 2
    hop = int((tp[1] - tp[0]) * fs)
    wlen = hop * 2 - 1
    win = hann(wlen + 2)[1:-1]
    frames = np.arange(2, n_frames - 1)
    if len(frames) == 0:
        return y
    origin = (frames - 1) * hop - (hop - 1)
    gidx = np.minimum(len(y), origin[:, None] + np.arange(wlen)[None, :])
    seg = excitation[gidx - 1] * win[None, :]
    mp = min_phase_spectrum(spectrogram[:, frames - 1].T, nfft)
    resp = sp_ifft(mp * sp_fft(seg, nfft, axis=1), axis=1).real
    for r in range(len(frames)):
        tgt = np.minimum(len(y), np.arange(origin[r], origin[r] + nfft)) - 1
        y[tgt] += resp[r]
    return y


def synthesis_requiem_np(f0, vuv, temporal_positions, spectrogram, band_ap_db, fs, seeds, cursor=None):
    """world/synthesisRequiem.py:12-25.  Returns (y, cursor)."""
    exc, cursor = requiem_excitation(temporal_positions, fs, f0, vuv, seeds["pulse"], seeds["noise"], band_ap_db, cursor)
    y = requiem_filter(exc, spectrogram, temporal_positions, len(np.asarray(f0)), fs)
    return y, cursor
