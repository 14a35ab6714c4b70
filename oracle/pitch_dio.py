"""ORACLE (test infrastructure) — DIO F0 estimator and StoneMask refinement.

Restates world/dio.py:10-476 and world/stonemask.py:8-76.  Only tests/, smoke() and bench.py's
cpu_baseline may import this.
"""
import math

import numpy as np
from scipy.signal import lfilter
from scipy.signal.windows import hann

from . import common as C

# (a0, a1, a2, b0, b1) of the reference's hard-coded decimation low-pass, world/dio.py:365-436.
# They are scipy.signal.cheby1(3, 0.05, 0.8/r) with the feedback signs flipped; the literal
# digits are data of the reference and are kept verbatim so that the recurrence is identical.
DECIMATE_COEFFS = {
    2: (0.041156734567757189, -0.42599112459189636, 0.041037215479961225, 0.16797464681802227, 0.50392394045406674),
    3: (0.95039378983237421, -0.67429146741526791, 0.15412211621346475, 0.071221945171178636, 0.21366583551353591),
    4: (1.4499664446880227, -0.98943497080950582, 0.24578252340690215, 0.036710750339322612, 0.11013225101796784),
    5: (1.7610939654280557, -1.2554914843859768, 0.3237186507788215, 0.021334858522387423, 0.06400457556716227),
    6: (1.9715352749512141, -1.4686795689225347, 0.3893908434965701, 0.013469181309343825, 0.040407543928031475),
    7: (2.1225239019534703, -1.6395144861046302, 0.44469707800587366, 0.0090366882681608418, 0.027110064804482525),
    8: (2.2357462340187593, -1.7780899984041358, 0.49152555365968692, 0.0063522763407111993, 0.019056829022133598),
    9: (2.3236003491759578, -1.8921545617463598, 0.53148928133729068, 0.0046331164041389372, 0.013899349212416812),
    10: (2.3936475118069387, -1.9873904075111861, 0.5658879979027055, 0.0034818622251927556, 0.010445586675578267),
    11: (2.450743295230728, -2.06794904601978, 0.59574774438332101, 0.0026822508007163792, 0.0080467524021491377),
    12: (2.4981398605924205, -2.1368928194784025, 0.62187513816221485, 0.0021097275904709001, 0.0063291827714127002),
}


def _iir3(sig: np.ndarray, r: int) -> np.ndarray:
    """world/dio.py:437-446: w[n]=x[n]+a0w[n-1]+a1w[n-2]+a2w[n-3]; y=b0w[n]+b1w[n-1]+b1w[n-2]+b0w[n-3],
    zero initial state; unknown r → all-zero filter."""
    a0, a1, a2, b0, b1 = DECIMATE_COEFFS.get(r, (0.0, 0.0, 0.0, 0.0, 0.0))
    return lfilter([b0, b1, b1, b0], [1.0, -a0, -a1, -a2], sig)


def decimate_by(x: np.ndarray, r: int) -> np.ndarray:
    """world/dio.py:451-476: mirror-pad 9, filter forward, reverse, filter, reverse, pick every r-th."""
    pad = 9
    n = len(x)
    head = 2 * x[0] - x[pad:0:-1]
    tail = 2 * x[-1] - x[n - 2 : n - 2 - pad : -1]
    buf = np.concatenate([head, x, tail])
    buf = _iir3(buf, r)[::-1]
    buf = _iir3(buf, r)[::-1]
    nout = np.ceil(n / r + 1)
    nbeg = int(r - r * nout + n)
    return buf[np.arange(nbeg, n + pad, r) + pad - 1].copy()


def lowcut_spectrum(y: np.ndarray, fs: float, lowest_f0: float) -> np.ndarray:
    """world/dio.py:74-88."""
    nfft = 2 ** math.ceil(math.log(len(y) + int(fs / lowest_f0 / 2 + 0.5) * 4, 2))
    cut = int(fs / 50 + 0.5)
    h = hann(2 * cut + 3)[1:-1]
    h = -h / np.sum(h)
    h[cut] += 1
    hz = np.zeros(nfft)
    hz[: cut + 1] = h[cut:]
    hz[nfft - cut :] = h[:cut]
    return np.fft.fft(y, nfft) * np.fft.fft(hz, nfft)


def crossing_intervals(sig: np.ndarray, fs: float):
    """Negative-going zero crossings → (interval midpoints [s], interval f0 [Hz]).
    world/dio.py:190-204 (1-based sample positions)."""
    nxt = np.empty_like(sig)
    nxt[:-1] = sig[1:]
    nxt[-1] = sig[-1]
    k = np.nonzero((nxt * sig < 0) & (nxt < sig))[0] + 1  # 1-based edge sample numbers
    fine = k - sig[k - 1] / (sig[k] - sig[k - 1])
    return (fine[:-1] + fine[1:]) / 2 / fs, fs / np.diff(fine)


def four_event_f0(filtered: np.ndarray, fs: float, times: np.ndarray, want_dev: bool):
    """world/dio.py:137-185 / world/harvest.py:262-269,499-529: four crossing trains
    (signal ±, first difference ±) interpolated to ``times``; mean (and ddof=1 std)."""
    d = np.diff(filtered)
    trains = [crossing_intervals(filtered, fs), crossing_intervals(-filtered, fs),
              crossing_intervals(d, fs), crossing_intervals(-d, fs)]
    usable = 1
    for loc, _ in trains:
        usable *= max(0, len(loc) - 2)
    if usable <= 0:
        return times * 0, (times * 0 + 1000 if want_dev else None)
    vals = np.stack([C.lerp_extrap(loc, f, times) for loc, f in trains])
    mean = np.mean(vals, axis=0)
    return mean, (np.std(vals, axis=0, ddof=1) if want_dev else None)


def dio_band_tables(boundary_f0_list, fs):
    """Per-band Nuttall low-pass taps and the delay index (argmax). world/dio.py:129-131."""
    out = []
    for bf in boundary_f0_list:
        half = int(fs / bf / 2 + 0.5)
        taps = C.nuttall_window(half * 4)
        out.append((taps, int(taps.argmax())))
    return out


def dio_np(x, fs, f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000, frame_period=5,
           allowed_range=0.1, index_bias_override=None):
    """world/dio.py:10-55 → dict(f0, f0_candidates, raw_f0_candidates, temporal_positions, vuv)."""
    x = np.asarray(x, dtype=np.float64)
    nf = C.frame_count(len(x), fs, frame_period)
    tp = C.frame_times(nf, frame_period)
    bands = np.arange(math.ceil(np.log2(f0_ceil / f0_floor) * channels_in_octave)) + 1
    bands = f0_floor * (2.0 ** (bands / channels_in_octave))
    y = decimate_by(x, int(fs / target_fs))
    fs_d = target_fs  # the true ratio is ignored
    spec = lowcut_spectrum(y, fs_d, f0_floor)
    tables = dio_band_tables(bands, fs_d)
    raw = np.zeros((len(bands), nf))
    stab = np.zeros((len(bands), nf))
    for b, bf in enumerate(bands):
        taps, bias = tables[b]
        if index_bias_override is not None:
            bias = int(index_bias_override[b])
        filt = np.real(np.fft.ifft(np.fft.fft(taps, len(spec)) * spec))
        filt = filt[bias + np.arange(1, len(y) + 1)]
        cand, dev = four_event_f0(filt, fs_d, tp, True)
        cand = np.array(cand, copy=True)
        dev = np.array(dev, copy=True)
        cand[(cand > bf) | (cand < bf / 2) | (cand > f0_ceil) | (cand < f0_floor)] = 0
        dev[cand == 0] = 100000
        raw[b] = cand
        stab[b] = np.exp(-(dev / np.maximum(cand, 0.0000001)))
    order = np.argsort(-stab, axis=0, kind="stable")
    cands = np.take_along_axis(raw, order, axis=0)
    kept = cands.copy()
    f0, vuv = dio_contour(cands, frame_period, f0_floor, allowed_range)
    return {"f0": f0, "f0_candidates": kept, "raw_f0_candidates": raw, "temporal_positions": tp, "vuv": vuv}


def _round6(v: np.ndarray) -> np.ndarray:
    """float('%.6f' % v) — world/dio.py:243."""
    return np.array([float("%.6f" % e) for e in v])


def _nearest_candidate(cur, past, cands, allowed_range):
    """world/dio.py:297-310."""
    ref = (cur * 3 - past) / 2
    best = cands[int(np.argmin(np.abs(ref - cands)))]
    if abs(1 - best / (ref + C.EPS)) > allowed_range:
        return 0.0
    return best


def _voiced_runs(f0):
    """world/dio.py:314-326 → list of (first, last) frame indices (inclusive) with the reference's
    boundary conventions (a run touching frame 0 starts at 1; the last boundary is len-2)."""
    v = (f0 != 0).astype(np.float64)
    dv = np.diff(v)
    bl = np.concatenate([[0], np.nonzero(dv != 0)[0], [len(v) - 2]]).astype(np.int64)
    first = math.ceil(-0.5 * dv[bl[1]])
    count = int(math.floor((len(bl) - (1 - first)) / 2))
    runs = []
    for i in range(count):
        runs.append((1 + bl[int((i - 1) * 2 + 1 + (1 - first)) + 1], bl[int(i * 2 + (1 - first)) + 1]))
    return runs


def dio_contour(cands, frame_period, f0_floor, allowed_range):
    """world/dio.py:216-293 — the four-step contour fix.  Mutates ``cands[0]`` ends like the
    reference does through its view."""
    vrm = int(1 / (frame_period / 1000) / f0_floor + 0.5) * 2 + 1
    base = cands[0]
    base[:vrm] = 0
    base[-vrm:] = 0
    n = len(base)
    # step 1: rapid change → 0 (compared on 6-decimal rounded values)
    s1 = base.copy()
    rb = _round6(base)
    i = np.arange(vrm - 1, n)
    jump = np.abs((rb[i] - rb[i - 1]) / (0.000001 + rb[i])) > allowed_range
    s1[i[jump]] = 0
    # step 2: erode by (vrm-1)/2 on both sides
    s2 = s1.copy()
    hw = int((vrm - 1) / 2)
    zero = (s1 == 0).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(zero)])
    centre = np.arange(hw, n - hw)
    has_zero = (csum[centre + hw + 1] - csum[centre - hw]) > 0
    s2[centre[has_zero]] = 0
    runs = _voiced_runs(s2)
    # step 3: extend each run forward
    s3 = s2.copy()
    for r, (st, ed) in enumerate(runs):
        limit = n - 1 if r == len(runs) - 1 else runs[r + 1][0] + 1
        for j in range(int(ed), int(limit)):
            s3[j + 1] = _nearest_candidate(s3[j], s3[j - 1], cands[:, j + 1], allowed_range)
            if s3[j + 1] == 0:
                break
    # step 4: extend each run backward
    s4 = s3.copy()
    for r in range(len(runs) - 1, -1, -1):
        limit = 1 if r == 0 else runs[r - 1][1]
        for j in range(int(runs[r][0]), int(limit) - 1, -1):
            s4[j - 1] = _nearest_candidate(s4[j], s4[j + 1], cands[:, j - 1], allowed_range)
            if s4[j - 1] == 0:
                break
    vuv = (s4 != 0).astype(np.float64)
    return s4, vuv


# ----------------------------------------------------------------------------------------------
# StoneMask
# ----------------------------------------------------------------------------------------------

def quantised_time_table(fs: float, kmax: int) -> np.ndarray:
    """table[k+kmax] = float('%.4f' % (k/fs)) — world/stonemask.py:38."""
    return np.array([float("{0:.4f}".format(e)) for e in (np.arange(-kmax, kmax + 1) / fs)])


def _weighted_if(inst_freq, power, f0, nfft, fs, harmonics):
    """world/stonemask.py:57-62: bins int(f0*nfft/fs*k+0.5)+1 (1-based), amplitude-weighted mean."""
    k = np.asarray(harmonics, dtype=np.float64)
    bins = C.half_up(f0[:, None] * nfft / fs * k[None, :]) + 1
    bins = bins.astype(np.int64) - 1
    fix = np.take_along_axis(inst_freq, bins, axis=1)
    amp = np.sqrt(np.take_along_axis(power, bins, axis=1))
    return np.sum(amp * fix, axis=1) / np.sum(amp * k[None, :], axis=1)


def stonemask_np(x, fs, temporal_positions, f0):
    """world/stonemask.py:8-76, frames grouped by FFT size."""
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(f0, dtype=np.float64)
    tp = np.asarray(temporal_positions, dtype=np.float64)
    out = np.copy(f0)
    act = np.nonzero(f0 != 0)[0]
    if len(act) == 0:
        return out
    hwl = np.ceil(3 * fs / f0[act] / 2)
    nfft_all = np.array([2 ** math.ceil(math.log(h * 2 + 1, 2) + 1) for h in hwl], dtype=np.int64)
    kmax = int(hwl.max())
    qt = quantised_time_table(fs, kmax)
    for nfft in np.unique(nfft_all):
        sel = np.nonzero(nfft_all == nfft)[0]
        rows = act[sel]
        h = hwl[sel].astype(np.int64)
        f0r = f0[rows]
        t0 = tp[rows]
        lmax = int(2 * h.max() + 1)
        j = np.arange(lmax)[None, :]
        valid = j < (2 * h[:, None] + 1)
        k = np.where(valid, j - h[:, None], 0)
        base_time = qt[k + kmax]
        idx_raw = C.half_up((t0[:, None] + base_time) * fs)
        win_t = (idx_raw - 1) / fs - t0[:, None]
        wlen = ((2 * h + 1) / fs)[:, None]
        main = 0.42 + 0.5 * np.cos(2 * math.pi * win_t / wlen) + 0.08 * np.cos(4 * math.pi * win_t / wlen)
        main = np.where(valid, main, 0.0)
        # derivative window: -(diff([0,w]) + diff([w,0]))/2 over the *valid* span
        prev = np.concatenate([np.zeros((len(rows), 1)), main[:, :-1]], axis=1)
        nxt = np.concatenate([main[:, 1:], np.zeros((len(rows), 1))], axis=1)
        dwin = np.where(valid, -((main - prev) + (nxt - main)) / 2, 0.0)
        idx = np.maximum(1, np.minimum(len(x), idx_raw)).astype(np.int64)
        seg = x[idx - 1]
        spec = np.fft.fft(seg * main, int(nfft), axis=1)
        dspec = np.fft.fft(seg * dwin, int(nfft), axis=1)
        num = spec.real * dspec.imag - spec.imag * dspec.real
        power = np.abs(spec) ** 2
        power[power == 0] = C.EPS
        fx = np.arange(nfft) / nfft * fs
        inst = fx[None, :] + num / power * fs / 2 / math.pi
        f_first = _weighted_if(inst, power, f0r, nfft, fs, [1, 2])
        neg = f_first < 0
        f_safe = np.where(neg, f0r, f_first)
        refined = _weighted_if(inst, power, f_safe, nfft, fs, [1, 2, 3, 4, 5, 6])
        refined = np.where(neg, 0.0, refined)
        bad = np.abs(refined - f0r) / f0r > 0.2
        out[rows] = np.where(bad, f0r, refined)
    return out
