"""ORACLE (test infrastructure, not product code) — shared numerical helpers.

CPU/NumPy restatement of pieces of tuanad121/Python-WORLD used only as the parity checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
(python-world_amd/) never imports this package.

Pinned against fixtures generated from the real reference (tests/golden/make_golden.py).
Citations are file:line into /root/reference.
"""
import math

import numpy as np

EPS = 2.220446049250313e-16


def frame_count(n_samples: int, fs: float, frame_period: float) -> int:
    """Number of analysis frames — world/dio.py:28, world/harvest.py:20,46."""
    return int(1000 * n_samples / fs / frame_period + 1)


def frame_times(n_frames: int, frame_period: float) -> np.ndarray:
    """world/dio.py:29."""
    return np.arange(0, n_frames) * frame_period / 1000


def nuttall_window(n) -> np.ndarray:
    """4-term Nuttall window — world/dio.py:208-212, world/d4c.py:237-245, world/harvest.py:563-567.

    ``n`` may be a float (d4c passes np.floor(...) results).  Evaluated through the same
    (1x4)@(4xN) product as the reference so that the even-N argmax tie falls
    the same way.
    """
    t = np.asmatrix(np.arange(n) * 2 * math.pi / (n - 1))
    coefs = np.array([0.355768, -0.487396, 0.144232, -0.012604])
    w = coefs @ np.cos(np.matrix([0, 1, 2, 3]).T @ t)
    return np.squeeze(np.asarray(w))


def half_up(v: np.ndarray) -> np.ndarray:
    """The reference's ``round_matlab`` — it only offsets by ±0.5, the caller truncates
    (world/cheaptrick.py:161-172)."""
    v = np.asarray(v, dtype=np.float64)
    return np.where(v > 0, v + 0.5, v - 0.5)


def lerp_extrap(xp: np.ndarray, fp: np.ndarray, xq: np.ndarray) -> np.ndarray:
    """scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate') for sorted ``xp``.

    Same arithmetic as SciPy's linear kernel: slope = (y_hi-y_lo)/(x_hi-x_lo);
    y = slope*(x-x_lo)+y_lo with the bracketing pair clipped to the end segments.
    ``fp`` may be (n,) or (n, m) (interpolation along axis 0).
    """
    xp = np.asarray(xp, dtype=np.float64)
    fp = np.asarray(fp, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    hi = np.clip(np.searchsorted(xp, xq), 1, len(xp) - 1)
    lo = hi - 1
    x_lo = xp[lo]
    x_hi = xp[hi]
    y_lo = fp[lo]
    y_hi = fp[hi]
    if fp.ndim == 1:
        slope = (y_hi - y_lo) / (x_hi - x_lo)
        return slope * (xq - x_lo) + y_lo
    slope = (y_hi - y_lo) / (x_hi - x_lo)[:, None]
    return slope * (xq - x_lo)[:, None] + y_lo


def lerp_extrap_unsorted(xp, fp, xq):
    """interp1d sorts its nodes first (assume_sorted=False)."""
    order = np.argsort(xp, kind="mergesort")
    return lerp_extrap(np.asarray(xp)[order], np.asarray(fp)[order], xq)


def gather_frames(x: np.ndarray, fs: float, f0: np.ndarray, pos: np.ndarray, half_length: float):
    """Left-aligned pitch-synchronous sample gather shared by CheapTrick and D4C.

    world/cheaptrick.py:86-94, world/d4c.py:95-101: half window = int(half_length*fs/f0+0.5);
    1-based centre int(pos*fs+0.501)+1; indices clamped to [1, len(x)].

    Returns (segment (F,Lmax), rel (F,Lmax) = sample offset from the centre, valid mask, hwl (F,)).
    """
    f0 = np.asarray(f0, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    hwl = np.floor(half_length * fs / f0 + 0.5).astype(np.int64)
    lmax = int(2 * hwl.max() + 1)
    j = np.arange(lmax)[None, :]
    rel = j - hwl[:, None]
    valid = j < (2 * hwl[:, None] + 1)
    centre = np.trunc(pos * fs + 0.501).astype(np.int64) + 1  # int() truncates toward zero
    idx = np.clip(centre[:, None] + rel, 1, len(x))
    seg = np.where(valid, x[idx - 1], 0.0)
    return seg, rel, valid, hwl


def remove_dc(seg: np.ndarray, window: np.ndarray, valid: np.ndarray, hwl: np.ndarray) -> np.ndarray:
    """waveform = s*w - w*mean(s*w)/mean(w) — world/cheaptrick.py:98, world/d4c.py:109."""
    length = (2 * hwl + 1).astype(np.float64)
    sw = seg * window
    m_sw = sw.sum(axis=1) / length
    m_w = window.sum(axis=1) / length
    return np.where(valid, sw - window * (m_sw / m_w)[:, None], 0.0)


def cumsum_band_mean(spec_full: np.ndarray, fs: float, fft_size: int, width: np.ndarray) -> np.ndarray:
    """Rectangular smoothing of width ``width`` (Hz, per row) via the doubled-spectrum cumsum
    and two linear look-ups — world/cheaptrick.py:103-131 (width 2f0/3), world/d4c.py:178-233.

    spec_full: (F, fft_size) Hermitian-symmetric rows.  Returns (F, fft_size//2+1) of
    (high-low) *not yet divided* by the width (callers scale differently).
    """
    nfft = fft_size
    axis2 = np.arange(2 * nfft) / nfft * fs - fs
    grid = axis2 + fs / nfft / 2
    seg = np.cumsum(np.concatenate([spec_full, spec_full], axis=1) * (fs / nfft), axis=1)
    centre = np.arange(nfft // 2 + 1) / nfft * fs
    half = np.asarray(width, dtype=np.float64)[:, None] / 2

    def lookup(xi):
        dx = grid[1] - grid[0]
        xi = np.maximum(grid[0], np.minimum(grid[-1], xi))
        base = np.floor((xi - grid[0]) / dx)
        frac = (xi - grid[0]) / dx - base
        b = base.astype(np.int64)
        dy = np.concatenate([np.diff(seg, axis=1), np.zeros((seg.shape[0], 1))], axis=1)
        return np.take_along_axis(seg, b, axis=1) + np.take_along_axis(dy, b, axis=1) * frac

    low = lookup(centre[None, :] - half)
    high = lookup(centre[None, :] + half)
    return high - low


def mirror_half(half_spec: np.ndarray) -> np.ndarray:
    """(F, K) → (F, 2(K-1)) even extension: [s, s[-2:0:-1]]."""
    return np.concatenate([half_spec, half_spec[:, -2:0:-1]], axis=1)


def low_band_replica(spec: np.ndarray, fs: float, fft_size: int, f0: np.ndarray, reach: np.ndarray) -> np.ndarray:
    """Mirror-add the bins below f0 around f0 and re-impose Hermitian symmetry, in place on a copy.

    world/cheaptrick.py:67-74 (reach = f0 + fs/fft) and world/d4c.py:213-222 (reach = 1.2*f0):
    nodes f0 - f_k for the bins with f_k < reach, linear inter/extrapolation back onto f_k,
    added to the bins with f_k < f0.
    """
    out = np.array(spec, dtype=np.float64, copy=True)
    axis = np.arange(fft_size) / fft_size * fs
    for i in range(out.shape[0]):
        sel = axis < reach[i]
        low_axis = axis[sel]
        rep = lerp_extrap_unsorted(f0[i] - low_axis, out[i, sel], low_axis)
        below = axis < f0[i]
        out[i, below] = rep[axis[: len(rep)] < f0[i]] + out[i, below]
    out[:, -1 : fft_size // 2 : -1] = out[:, 1 : fft_size // 2]
    return out
