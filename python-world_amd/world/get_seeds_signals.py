"""Seed signals for Requiem synthesis — drop-in for world/get_seeds_signals.py:8 of the reference.

Host Python by design: a few kilobytes of per-sampling-rate constant tables built
once and uploaded.  The two global random streams (`random`, `numpy.random`) are consumed in the same
order as the reference, so a seeded call yields the same tables.
"""
import random

import numpy as np
from scipy.fft import fft, ifft
from scipy.signal.windows import hann

_BAND_STEP = 3000  # Hz between raised-cosine band centres
_UPPER = 15000


def _shuffled_signs(count):
    """±2 pool, first half positive, shuffled by `count` random swaps (world/get_seeds_signals.py:61-69)."""
    pool = np.full(count, 2.0)
    pool[count // 2:] = -2.0
    for i in range(count):
        j = random.randint(0, count - 1)
        pool[i], pool[j] = pool[j], pool[i]
    return pool


def _velvet_segment(length):
    """One short velvet-noise segment: one ±2 impulse per 4-sample cell at a random offset."""
    cell = 4
    cells = int(length // cell + 0.5)
    seg = np.zeros(length)
    signs = _shuffled_signs(cells)
    seg[cell * np.arange(cells) + np.random.randint(cell, size=cells)] = signs
    return seg


def _modified_velvet_noise(n, fs):
    """Concatenation of randomly chosen short segments (world/get_seeds_signals.py:40-53).  The segment
    lengths come out as int(8*(p*fs/48000 + 0.5)) because the reference's round helper only offsets
    (SURVEY Q1): 25/84/164 samples at 16 kHz."""
    lengths = [int(8 * (p * fs / 48000 + 0.5)) for p in (8, 30, 60)]
    out = np.zeros(n + max(lengths) + 1)
    at = 0
    while True:
        ln = lengths[random.randint(0, len(lengths) - 1)]
        out[at:at + ln] = _velvet_segment(ln)
        at += ln
        if at >= n - 1:
            return out[:n]


def get_seeds_signals(fs: int, fft_size: int = None, noise_length: int = None):
    """{'pulse': (fft_size, nb), 'noise': (noise_length, nb)} with nb = 2 + floor(min(15000, fs/2-3000)/3000)."""
    if fft_size is None:
        fft_size = int(1024 * (2 ** np.ceil(np.log2(fs / 48000))))
    if noise_length is None:
        noise_length = int(2 ** np.ceil(np.log2(fs / 2)))
    freq = np.arange(fft_size // 2 + 1) * fs / fft_size
    nb = int(2 + np.floor(min(_UPPER, fs / 2 - _BAND_STEP) / _BAND_STEP))
    pulse = np.zeros((fft_size, nb))
    noise = np.zeros((noise_length, nb))
    velvet_spec = fft(_modified_velvet_noise(noise_length, fs), noise_length)
    for b in range(nb):
        shape = 0.5 + 0.5 * np.cos(((freq - (_BAND_STEP * b)) / (_BAND_STEP * 2)) * 2 * np.pi)
        shape[freq > (_BAND_STEP * (b + 1))] = 0
        shape[freq < (_BAND_STEP * (b - 1))] = 0
        if b == nb - 1:
            shape[freq > (_BAND_STEP * b)] = 1  # the top band is a high-pass
        pulse[:, b] = np.fft.fftshift(ifft(np.r_[shape, shape[-2:0:-1]]).real)
        noise[:, b] = ifft(velvet_spec * fft(pulse[:, b], noise_length)).real
    window = hann(fft_size + 2)[1:-1]
    pulse[:, 0] = pulse[:, 0] - np.mean(pulse[:, 0]) * window / np.mean(window)  # DC-free lowest band
    return {'pulse': pulse, 'noise': noise}


def get_seeds_signals_device(fs, seed=0, fft_size=None, noise_length=None, device_index=None, want_velvet=False, rt=None):
    """The same tables generated ON the device (wh_requiem_seeds): {'pulse_d', 'noise_d'} torch tensors that
    WorldBatch.decode_device(..., seeds=...) uses in place — no host RNG, no upload.  The pulses are the exact
    deterministic ones; the velvet noise follows the reference's construction with a counter-based Philox stream
    ``seed`` (statistically equivalent, not sample-identical: tests/test_hip_seeds.py)."""
    import ctypes

    from . import _hip

    if rt is None:  # ``rt``: the runtime (context + stream) to generate on; default: lane 0 of ``device_index``
        rt = _hip.Runtime.get(device_index)
    if fft_size is None:
        fft_size = int(1024 * (2 ** np.ceil(np.log2(fs / 48000))))
    if noise_length is None:
        noise_length = int(2 ** np.ceil(np.log2(fs / 2)))
    nb = int(2 + np.floor(min(_UPPER, fs / 2 - _BAND_STEP) / _BAND_STEP))
    pulse_d = rt.empty((int(fft_size), nb))
    noise_d = rt.empty((int(noise_length), nb))
    velvet_d = rt.empty((int(noise_length),)) if want_velvet else None
    _hip.check(rt.lib.wh_requiem_seeds(rt.ctx, rt.stream(), float(fs), int(fft_size), int(noise_length), nb,
                                       ctypes.c_uint64(int(seed)), rt.ptr(pulse_d), rt.ptr(noise_d), rt.ptr(velvet_d)))
    out = {'pulse_d': pulse_d, 'noise_d': noise_d}
    if want_velvet:
        out['velvet_d'] = velvet_d
    return out
