"""ORACLE (test infrastructure) — CheapTrick spectral envelope, all frames at once.

Restates world/cheaptrick.py:9-157 of the reference with the per-frame loop turned into
row-wise NumPy over an (F, Lmax) gather.  Only tests/, smoke() and bench.py's cpu_baseline
may import this.
"""
import numpy as np

from . import common as C


def default_fft_size(fs: float, f0_floor: float = 71.0) -> int:
    """world/cheaptrick.py:22."""
    return int(2 ** np.ceil(np.log2(3 * fs / f0_floor + 1)))


def cheaptrick_np(x, fs, f0, vuv, temporal_positions, q1=-0.15, fft_size=None, want_ps=True):
    """Returns (spectrogram (K,F), ps_spectrogram (fft,F) complex or None, f0_used (F,)).

    ``f0_used`` is the f0 vector after the 500 Hz substitutions the reference writes back into the
    caller's array (world/cheaptrick.py:26-27,32-33).
    """
    x = np.asarray(x, dtype=np.float64)
    if fft_size is None:
        fft_size = default_fft_size(fs)
    fft_size = int(fft_size)
    low_limit = fs * 3.0 / (fft_size - 3.0)
    f0u = np.array(f0, dtype=np.float64, copy=True)
    f0u[np.asarray(vuv) == 0] = 500.0
    f0u[f0u < low_limit] = 500.0
    pos = np.asarray(temporal_positions, dtype=np.float64)

    # step 1: 3*T0 Hann window, L2-normalised, DC removed (cheaptrick.py:79-99)
    seg, rel, valid, hwl = C.gather_frames(x, fs, f0u, pos, 1.5)
    window = np.where(valid, 0.5 * np.cos(np.pi * (rel / fs / 1.5) * f0u[:, None]) + 0.5, 0.0)
    window = window / np.sqrt(np.sum(window ** 2, axis=1))[:, None]
    wave = C.remove_dc(seg, window, valid, hwl)

    # power spectrum + replica below f0 (cheaptrick.py:64-75); np.fft crops rows longer than n (Q7)
    ps = np.fft.fft(wave, fft_size, axis=1)
    power = np.abs(ps) ** 2
    power = C.low_band_replica(power, fs, fft_size, f0u, f0u + fs / fft_size)

    # step 2: rectangular smoothing of width 2*f0/3 (cheaptrick.py:103-118).  The reference adds an unseeded
    # rand*eps dither so that log() never sees 0 (Q10); its mean eps/2 is added here (deterministic)
    smoothed = C.cumsum_band_mean(power, fs, fft_size, 2.0 * f0u / 3.0) * 1.5 / f0u[:, None] + 0.5 * C.EPS

    # step 3: liftering (cheaptrick.py:136-157)
    envelope = lifter_recover(C.mirror_half(smoothed), f0u, fs, fft_size, q1)
    return envelope.T.copy(), (ps.T.copy() if want_ps else None), f0u


def lifter_recover(sym_spec, f0, fs, fft_size, q1):
    half = fft_size // 2
    quef = np.arange(fft_size) / fs
    f0c = np.asarray(f0, dtype=np.float64)[:, None]
    smooth = np.empty((len(f0c), fft_size))
    smooth[:, 0] = 1.0
    arg = np.pi * f0c * quef[None, 1:]
    smooth[:, 1:] = np.sin(arg) / arg
    smooth[:, half + 1 :] = smooth[:, half - 1 : 0 : -1]
    comp = (1 - 2 * q1) + 2 * q1 * np.cos(2 * np.pi * quef[None, :] * f0c)
    comp[:, half + 1 :] = comp[:, half - 1 : 0 : -1]
    ceps = np.fft.fft(np.log(sym_spec), axis=1)
    env = np.exp(np.real(np.fft.ifft(ceps * smooth * comp, axis=1)))
    return env[:, : half + 1]
