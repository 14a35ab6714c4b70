"""SWIPE' F0 estimator — drop-in for world/swipe.py:9 of the reference (f0_method='swipe' of World.encode,
world/main.py:45-46,134-135), executed by the HIP kernels behind wh_swipe (include/world_hip.h).

The estimator is defined by a handful of tables — candidate pitches, the power-of-two window sizes with their Hann
windows, the cubic-spline resampling of each window size's magnitude bins onto the ERB grid (a fixed matrix: SciPy's
interp1d(kind='cubic') is linear in its data), one kernel per candidate, the parabolic-refinement grids.  They are
built here with the reference's own NumPy / SciPy expressions and handed through the ABI as data; the per-frame
work (STFT, the two dense products on the FP64 matrix cores, interpolation onto the output grid, peak picking) runs
on the device."""
import ctypes
import functools
from decimal import ROUND_HALF_UP, Decimal

import numpy as np

from . import _hip

_FINE = 20  # refinement grid slots per candidate (the 1/768-octave grid over two candidate steps has 17-18 points)


class _Window(ctypes.Structure):
    _fields_ = [("ws", ctypes.c_int32), ("hop", ctypes.c_int32), ("j0", ctypes.c_int32), ("n_c", ctypes.c_int32),
                ("h_window", ctypes.c_void_p), ("h_interp", ctypes.c_void_p), ("h_kernels", ctypes.c_void_p),
                ("h_mu", ctypes.c_void_p), ("table_tag", ctypes.c_uint64)]


def _round_half_up(v):
    return int(Decimal(float(v)).quantize(0, ROUND_HALF_UP))


def _hz2erbs(hz):
    return 21.4 * np.log10(1 + hz / 229)


def _erbs2hz(erbs):
    return (10 ** (erbs / 21.4) - 1) * 229


def _sieve_harmonics(n):
    """Harmonic numbers > 1 of a candidate's kernel = what the reference's `sieve(n)` returns (world/swipe.py:158-172,
    called at :130).  That sieve strikes the multiples of each prime p only while p < sqrt(n) — strictly — so when n
    is itself the square of a prime (4, 9, 25, 49, 121, 169, 289, ...) nothing ever strikes n and it stays in the list
    next to the true primes.  The kernels are defined by that list, quirk included (19 of the 336 candidates at 16 kHz)."""
    flags = np.ones(max(n + 1, 2), dtype=bool)
    flags[:2] = False
    for p in range(2, int(n ** 0.5) + 1):
        if flags[p]:
            flags[p * p::p] = False
    out = [int(p) for p in np.nonzero(flags)[0] if p <= n]
    r = int(round(n ** 0.5)) if n >= 4 else 0
    if r * r == n and flags[r]:
        out.append(n)
    return out


def _kernel(f, pc):
    """K+-normalised kernel of one candidate on the ERB grid (world/swipe.py:127-146)."""
    n = int(np.fix(f[-1] / pc - 0.75))
    k = np.zeros(len(f))
    q = f / pc
    for i in [1] + _sieve_harmonics(n):
        a = np.abs(q - i)
        peak = a < 0.25
        k[peak] = np.cos(2 * np.pi * q[peak])
        valley = np.logical_and(0.25 < a, a < 0.75)
        k[valley] = k[valley] + np.cos(2 * np.pi * q[valley]) / 2
    k *= np.sqrt(1 / f)
    k /= np.linalg.norm(k[k > 0])
    return k


@functools.lru_cache(maxsize=8)
def swipe_tables(fs, plim_lo, plim_hi):
    """Everything wh_swipe needs for (fs, plim): see the module docstring.  Cached: ~10 MB at 16 kHz."""
    from scipy import interpolate

    plim = np.array([plim_lo, plim_hi], dtype=np.float64)
    log2pc = np.arange(np.log2(plim[0]) * 96, np.log2(plim[-1]) * 96) * (1 / 96)
    pc = 2 ** log2pc
    log_ws = [_round_half_up(e) for e in np.log2(4 * 2 * fs / plim)]
    ws = 2 ** np.arange(log_ws[0], log_ws[1] - 1, -1)
    p0 = 4 * 2 * fs / ws
    d = 1 + log2pc - np.log2(4 * 2 * fs / ws[0])
    f_erbs = _erbs2hz(np.arange(_hz2erbs(pc[0] / 4), _hz2erbs(fs / 2), 0.1))
    windows = []
    for i, w_size in enumerate(ws):
        w_size = int(w_size)
        dn = _round_half_up(4 * fs / p0[i])
        overlap = max(0, np.round(w_size - dn))
        hop = int(w_size - overlap)
        if i == len(ws) - 1:
            j = np.where(d - (i + 1) > -1)[0]
            k = np.where(d[j] - (i + 1) < 0)[0]
        elif i == 0:
            j = np.where(d - (i + 1) < 1)[0]
            k = np.where(d[j] - (i + 1) > 0)[0]
        else:
            j = np.where(np.abs(d - (i + 1)) < 1)[0]
            k = np.arange(len(j))
        assert len(j) and np.all(np.diff(j) == 1)  # a contiguous run of candidates
        mu = np.ones(len(j))
        mu[k] = 1 - np.abs(d[j[k]] - i - 1)
        f = np.arange(w_size // 2 + 1) * fs / w_size
        # interp1d(f, |X|.T, kind='cubic')(f_erbs) is linear in |X|: resample the identity once to get the matrix
        interp = interpolate.interp1d(f, np.eye(len(f)), kind="cubic", axis=0)(f_erbs)  # (n_erb, bins)
        kernels = np.stack([_kernel(f_erbs, pc[c]) for c in j])                        # (n_c, n_erb)
        windows.append({"ws": w_size, "hop": hop, "j0": int(j[0]), "n_c": len(j),
                        "window": np.ascontiguousarray(np.hanning(w_size + 2)[1:-1]),
                        "interp": np.ascontiguousarray(interp.T), "kernels": np.ascontiguousarray(kernels.T),
                        "mu": np.ascontiguousarray(mu)})
        # identity of the two matrices for the library's table cache: content hash, taken once per cached table set
        windows[-1]["tag"] = _hip.table_tag(windows[-1]["interp"], windows[-1]["kernels"])
    ntc = np.zeros((len(pc), 3))
    fine = np.zeros((len(pc), _FINE))
    n_fine = np.zeros(len(pc), dtype=np.int32)
    for i in range(1, len(pc) - 1):
        idx = np.arange(i - 1, i + 2)
        tc = 1 / pc[idx]
        ntc[i] = (tc / tc[1] - 1) * 2 * np.pi
        ftc = 1 / (2 ** np.arange(np.log2(pc[idx[0]]), np.log2(pc[idx[2]]) + 1 / 12 / 64, 1 / 12 / 64))
        grid = (ftc / tc[1] - 1) * 2 * np.pi
        assert len(grid) <= _FINE
        n_fine[i] = len(grid)
        fine[i, :len(grid)] = grid
    return {"pc": np.ascontiguousarray(pc), "n_erb": len(f_erbs), "windows": windows, "ntc": ntc, "fine": fine,
            "n_fine": n_fine}


def swipe_device(rt, batch, x_d, fs, plim=(71, 800), dt=0.005, sTHR=float("-inf")):
    """Device-resident core: (f0, vuv) device tensors on the batch's frame grid (which must be the dt grid)."""
    tb = swipe_tables(int(fs), float(plim[0]), float(plim[-1]))
    vp = ctypes.c_void_p
    wins = (_Window * len(tb["windows"]))()
    for s, w in zip(wins, tb["windows"]):
        s.ws, s.hop, s.j0, s.n_c = w["ws"], w["hop"], w["j0"], w["n_c"]
        s.h_window = w["window"].ctypes.data
        s.h_interp = w["interp"].ctypes.data
        s.h_kernels = w["kernels"].ctypes.data
        s.h_mu = w["mu"].ctypes.data
        s.table_tag = w["tag"]
    f0 = rt.empty((batch.total_frames,))
    vuv = rt.empty((batch.total_frames,))
    thr = float(sTHR) if np.isfinite(sTHR) else -1.0e308
    _hip.check(rt.lib.wh_swipe(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), float(fs), float(dt), thr,
                               len(tb["pc"]), tb["pc"].ctypes.data_as(vp), int(tb["n_erb"]), len(wins),
                               ctypes.cast(wins, vp), tb["ntc"].ctypes.data_as(vp), tb["fine"].ctypes.data_as(vp),
                               tb["n_fine"].ctypes.data_as(vp), _FINE, rt.ptr(f0), rt.ptr(vuv)))
    return f0, vuv


@_hip.serialised
def swipe(fs, x, plim=[71, 800], dt=0.005, sTHR=float('-inf')):
    """Same contract as the reference: {'temporal_positions', 'f0', 'vuv'} on the dt grid."""
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    num = int(1000 * len(x) / fs / (dt * 1000) + 1)
    t = np.arange(0, num) * dt
    batch = rt.make_batch([0, len(x)], [0, num])
    f0, vuv = swipe_device(rt, batch, rt.to_device(x), fs, plim, dt, sTHR)
    rt.check_flags("swipe")
    return {'temporal_positions': t, 'f0': f0.cpu().numpy(), 'vuv': vuv.cpu().numpy()}
