"""ORACLE (test infrastructure) — Harvest F0 estimator.

Restates world/harvest.py:17-609.  The (candidate row × 1 ms frame) refinement that the reference
farms out to a process pool (harvest.py:131-150) is evaluated here as grouped row-wise NumPy.
Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import math
from decimal import ROUND_HALF_UP, Decimal

import numpy as np
from scipy import signal
from scipy.fft import fft as sp_fft

from . import common as C
from .pitch_dio import four_event_f0


def downsample_8k(x: np.ndarray, fs: float, target_fs: float = 8000):
    """world/harvest.py:58-71,584-609: constant edge pad, zero-phase Chebyshev-I(3, 0.05 dB, 0.8/r)
    via filtfilt(padlen=9), keep every r-th sample, trim, remove the mean."""
    r = int(fs / target_fs + 0.5)
    if fs <= target_fs:
        y = np.array(x, dtype=np.float64, copy=True)
        fs_d = fs
    else:
        offset = int(np.ceil(140 / r) * r)
        xp = np.concatenate([np.ones(offset) * x[0], x, np.ones(offset) * x[-1]])
        b, a = signal.cheby1(3, 0.05, 0.8 / r)
        z = signal.filtfilt(b, a, xp, padlen=3 * (max(len(a), len(b)) - 1))
        nd = len(z)
        n_out = np.ceil(nd / r)
        n_beg = int(r - (r * n_out - nd))
        y0 = z[n_beg - 1 :: r]
        fs_d = fs / r
        y = y0[int(offset / r) : int(-offset / r)]
    y = y - np.mean(y)
    return y, fs_d


def harvest_bands(f0_floor, f0_ceil, channels_in_octave=40):
    """world/harvest.py:22-29."""
    lo = f0_floor * 0.9
    hi = f0_ceil * 1.1
    b = np.arange(np.ceil(np.log2(hi / lo) * channels_in_octave)) + 1
    b = 2.0 ** (b / channels_in_octave)
    return b * lo


def band_pass_taps(bf: float, fs_d: float):
    """world/harvest.py:253-256: Nuttall(2h+1) × cos carrier, h = round-half-up(2 fs_d / bf)."""
    h = int(Decimal(fs_d / bf * 2).quantize(0, ROUND_HALF_UP))
    taps = C.nuttall_window(h * 2 + 1) * np.cos(2 * math.pi * bf * np.arange(-h, h + 1) / fs_d)
    return taps, h


def raw_candidates(y, fs_d, fs, f0_floor, f0_ceil, times):
    """world/harvest.py:33-34,75-84,252-278 → (n_bands, F1)."""
    floor_adj = f0_floor * 0.9
    nfft = int(2 ** np.ceil(np.log2(len(y) + int(fs / floor_adj * 4 + 0.5) + 1)))
    spec = np.fft.fft(y, nfft)
    bands = harvest_bands(f0_floor, f0_ceil)
    out = np.zeros((len(bands), len(times)))
    for b, bf in enumerate(bands):
        taps, h = band_pass_taps(bf, fs_d)
        filt = np.real(np.fft.ifft(np.fft.fft(taps, nfft) * spec))
        filt = filt[(h + 1) + np.arange(len(y))]
        cand, _ = four_event_f0(filt, fs_d, times, False)
        cand = np.array(cand, copy=True)
        cand[(cand > bf * 1.1) | (cand < bf * 0.9) | (cand > f0_ceil) | (cand < f0_floor)] = 0
        out[b] = cand
    return out


def detect_candidates(raw):
    """world/harvest.py:88-110: runs of ≥10 adjacent live channels → their mean."""
    nch, nfr = raw.shape
    cands = np.zeros((int(nch / 10 + 0.5), nfr))
    live = (raw > 0).astype(np.int8)
    live[0] = 0
    live[-1] = 0
    edge = np.diff(live, axis=0)
    most = 0
    for i in range(nfr):
        st = np.nonzero(edge[:, i] == 1)[0]
        if len(st) == 0:
            continue
        ed = np.nonzero(edge[:, i] == -1)[0]
        c = 0
        for s, e in zip(st, ed):
            if e - s >= 10:
                cands[c, i] = np.mean(raw[s + 1 : e + 1, i])
                c += 1
        most = max(most, c)
    return cands, most


def overlap_candidates(cands, max_c):
    """world/harvest.py:114-125: copies shifted by -3…+3 frames (note the stray seeding of row 0)."""
    n = 3
    reps = 2 * n + 1
    nfr = cands.shape[1]
    out = np.zeros((reps * max_c, nfr))
    out[0, :] = cands[reps - 1, :]
    for i in range(reps):
        s = i - n
        rows = slice(i * max_c, (i + 1) * max_c)
        if s <= 0:
            out[rows, -s:nfr] = cands[:max_c, 0 : nfr + s]
        else:
            out[rows, 0 : nfr - s] = cands[:max_c, s:nfr]
    return out


def refine_pairs(y, fs_d, t, f0c, f0_floor, f0_ceil, chunk=20000):
    """GetRefinedF0 (world/harvest.py:169-211) for flat arrays of (time, candidate) pairs, all
    candidates non-zero.  Returns (refined, score)."""
    n = len(f0c)
    ref = np.zeros(n)
    score = np.zeros(n)
    hwl = np.ceil(3 * fs_d / f0c / 2)
    nfft_all = (2 ** np.ceil(np.log2(hwl * 2 + 1) + 1)).astype(np.int64)
    for nfft in np.unique(nfft_all):
        sel_all = np.nonzero(nfft_all == nfft)[0]
        for c0 in range(0, len(sel_all), chunk):
            sel = sel_all[c0 : c0 + chunk]
            h = hwl[sel].astype(np.int64)
            f0r = f0c[sel]
            t0 = t[sel]
            lmax = int(2 * h.max() + 1)
            j = np.arange(lmax)[None, :]
            ln = 2 * h[:, None] + 1
            valid = j < ln
            k = np.where(valid, j - h[:, None], 0)
            idx_raw = C.half_up((t0[:, None] + k / fs_d) * fs_d + 0.001)
            common = math.pi * ((idx_raw - 1) / fs_d - t0[:, None]) / (ln / fs_d)
            main = np.where(valid, 0.42 + 0.5 * np.cos(2 * common) + 0.08 * np.cos(4 * common), 0.0)
            prev = np.concatenate([np.zeros((len(sel), 1)), main[:, :-1]], axis=1)
            nxt = np.concatenate([main[:, 1:], np.zeros((len(sel), 1))], axis=1)
            # interior: -((w[i+1]-w[i]) + (w[i]-w[i-1]))/2 ; ends: -w[1]/2 and w[-2]/2
            dwin = -((nxt - main) + (main - prev)) / 2
            dwin[:, 0] = -main[:, 1] / 2
            last = (ln[:, 0] - 1)
            rows = np.arange(len(sel))
            dwin[rows, last] = main[rows, last - 1] / 2
            dwin = np.where(valid, dwin, 0.0)
            idx = (np.maximum(1, np.minimum(len(y), idx_raw)) - 1).astype(np.int64)
            seg = y[idx]
            sp = sp_fft(seg * main, int(nfft), axis=1)
            dsp = sp_fft(seg * dwin, int(nfft), axis=1)
            num = sp.real * dsp.imag - sp.imag * dsp.real
            power = np.abs(sp) ** 2
            with np.errstate(invalid="ignore", divide="ignore"):
                inst = (np.arange(nfft)[None, :] / nfft + num / power / 2 / math.pi) * fs_d
            nh = np.minimum(np.floor(fs_d / 2 / f0r), 6).astype(np.int64)
            harm = np.arange(1, 7)[None, :]
            hvalid = harm <= nh[:, None]
            bins = C.half_up(f0r[:, None] * nfft / fs_d * harm).astype(np.int64)
            bins = np.where(hvalid, bins, 0)
            il = np.take_along_axis(inst, bins, axis=1)
            amp = np.sqrt(np.take_along_axis(power, bins, axis=1))
            il0 = np.where(hvalid, il, 0.0)
            amp0 = np.where(hvalid, amp, 0.0)
            with np.errstate(invalid="ignore", divide="ignore"):
                rf = np.sum(amp0 * il0, axis=1) / np.sum(amp0 * harm, axis=1)
                var = np.where(hvalid, np.abs((il / harm - f0r[:, None]) / f0r[:, None]), 0.0)
                sc = 1 / (0.000000000001 + np.sum(var, axis=1) / nh)
            bad = (rf < f0_floor) | (rf > f0_ceil) | (sc < 2.5)
            ref[sel] = np.where(bad, 0.0, rf)
            score[sel] = np.where(bad, 0.0, sc)
    return ref, score


def refine_candidates(y, fs_d, times, cands, f0_floor, f0_ceil):
    """world/harvest.py:131-150."""
    out_f0 = np.zeros_like(cands)
    out_sc = np.zeros_like(cands)
    rr, cc = np.nonzero(cands)
    if len(rr):
        f, s = refine_pairs(y, fs_d, times[cc], cands[rr, cc], f0_floor, f0_ceil)
        out_f0[rr, cc] = f
        out_sc[rr, cc] = s
    return out_f0, out_sc


def prune_isolated(cands, scores, threshold=0.05):
    """world/harvest.py:215-248: keep a candidate only if some candidate in frame i-1 or i+1 is
    within 5 %; first/last frame untouched."""
    out_c = np.array(cands)
    out_s = np.array(scores)
    nfr = cands.shape[1]
    if nfr < 3:
        return out_c, out_s
    mid = cands[:, 1:-1]
    err = np.empty_like(mid)
    step = 2048
    with np.errstate(invalid="ignore", divide="ignore"):
        def nearest(m, neigh):
            e = np.abs(m[:, None, :] - neigh[None, :, :]) / m[:, None, :]
            e = np.where(e > 1, 1.0, e)  # SelectBestF0 starts from allowed_range = 1
            e = np.where(np.isnan(e), 1.0, e)
            return np.min(e, axis=1)
        for c0 in range(0, mid.shape[1], step):
            m = mid[:, c0 : c0 + step]
            err[:, c0 : c0 + step] = np.minimum(nearest(m, cands[:, 2 + c0 : 2 + c0 + m.shape[1]]),
                                                nearest(m, cands[:, c0 : c0 + m.shape[1]]))
    kill = (mid != 0) & (err > threshold)
    out_c[:, 1:-1][kill] = 0
    out_s[:, 1:-1][kill] = 0
    return out_c, out_s


def boundary_list(f0):
    """world/harvest.py:572-580: [first0,last0,first1,last1,…] of voiced runs, ends forced unvoiced."""
    v = (np.asarray(f0) != 0).astype(np.int64)
    v[0] = 0
    v[-1] = 0
    bl = np.nonzero(np.diff(v) != 0)[0]
    bl[0::2] += 1
    return bl


def _pick(ref, col, allowed):
    """SelectBestF0 — world/harvest.py:238-248 (later ties win; NaN skipped)."""
    best = 0.0
    best_err = allowed
    for c in col:
        e = abs(ref - c) / ref
        if e > best_err:
            continue
        best = c
        best_err = e
    return best


def _extend(f0, origin, last_point, shift, cands, allowed):
    """world/harvest.py:408-429."""
    ext = np.array(f0)
    cur = ext[origin]
    reached = origin
    miss = 0
    last_point += shift
    for i in range(origin, last_point, shift):
        ext[i + shift] = _pick(cur, cands[:, i + shift], allowed)
        if ext[i + shift] != 0:
            cur = ext[i + shift]
            miss = 0
            reached = i + shift
        else:
            miss += 1
        if miss == 4:
            break
    return ext, reached


def _score_of(f0, col, sc):
    """world/harvest.py:490-495."""
    s = 0
    for k in range(len(col)):
        if f0 == col[k] and s < sc[k]:
            s = sc[k]
    return s


def _merge(channels, rng, cands, scores):
    """world/harvest.py:442-486."""
    order = np.argsort(rng[:, 0], axis=0, kind="stable")
    f0 = channels[order[0]]
    rng = rng.astype(np.int64)
    o0 = order[0]
    for q in range(1, channels.shape[0]):
        oq = order[q]
        if rng[oq, 0] - rng[o0, 1] > 0:
            f0[rng[oq, 0] : rng[oq, 1] + 1] = channels[oq, rng[oq, 0] : rng[oq, 1] + 1]
            rng[o0, 0] = rng[oq, 0]
            rng[o0, 1] = rng[oq, 1]
        else:
            st1, ed1, st2, ed2 = int(rng[o0, 0]), int(rng[o0, 1]), int(rng[oq, 0]), int(rng[oq, 1])
            if st1 <= st2 and ed1 >= ed2:
                continue
            f2 = channels[oq]
            s1 = 0
            s2 = 0
            for i in range(st2, ed1 + 1):
                s1 = s1 + _score_of(f0[i], cands[:, i], scores[:, i])
                s2 = s2 + _score_of(f2[i], cands[:, i], scores[:, i])
            merged = f0.copy()
            if s1 > s2:
                merged[ed1 : ed2 + 1] = f2[ed1 : ed2 + 1]
            else:
                merged[st2 : ed2 + 1] = f2[st2 : ed2 + 1]
            f0 = merged
            rng[o0, 1] = ed2
    return f0


def fix_contour(cands, scores):
    """world/harvest.py:301-404 → (f0 (F1,), vuv (F1,))."""
    nfr = cands.shape[1]
    # base: best-scoring candidate per frame
    base = cands[np.argmax(scores, axis=0), np.arange(nfr)]
    # step 1: continuity against the linear prediction and the previous frame (0.8 %)
    s1 = base.copy()
    s1[0] = 0
    s1[1] = 0
    ar = 0.008
    with np.errstate(invalid="ignore", divide="ignore"):
        pred = base[1:-1] * 2 - base[:-2]
        cur = base[2:]
        prv = base[1:-1]
        drop = (cur != 0) & (np.abs((cur - pred) / (pred + C.EPS)) > ar) & (np.abs((cur - prv) / (prv + C.EPS)) > ar)
    s1[2:][drop] = 0
    # step 2: voiced runs shorter than 6 frames vanish
    s2 = s1.copy()
    bl = boundary_list(s1)
    for i in range(1, len(bl) // 2 + 1):
        if bl[2 * i - 1] - bl[2 * i - 2] < 6:
            s2[bl[2 * i - 2] : bl[2 * i - 1] + 1] = 0
    # step 3: extend every run both ways through the candidate map, keep long ones, merge
    s3 = np.array(s2)
    bl = boundary_list(s2)
    nsec = len(bl) // 2
    kept_f0 = []
    kept_rng = []
    for i in range(1, nsec + 1):
        ch = np.zeros(nfr)
        ch[bl[2 * i - 2] : bl[2 * i - 1] + 1] = s2[bl[2 * i - 2] : bl[2 * i - 1] + 1]
        ext, r1 = _extend(ch, int(bl[2 * i - 1]), int(min(nfr - 2, bl[2 * i - 1] + 100)), 1, cands, 0.18)
        seq, r0 = _extend(ext, int(bl[2 * i - 2]), int(max(1, bl[2 * i - 2] - 100)), -1, cands, 0.18)
        mean_f0 = np.mean(seq[int(r0) : int(r1) + 1])
        if 2200 / mean_f0 < r1 - r0:
            kept_f0.append(seq)
            kept_rng.append((r0, r1))
    if kept_f0:
        s3 = _merge(np.array(kept_f0), np.array(kept_rng, dtype=np.float64), cands, scores)
    # step 4: bridge unvoiced gaps shorter than 9 frames linearly
    s4 = s3.copy()
    bl = boundary_list(s3)
    for i in range(1, len(bl) // 2):
        dist = bl[2 * i] - bl[2 * i - 1] - 1
        if dist >= 9:
            continue
        a0 = s3[bl[2 * i - 1]] + 1
        a1 = s3[bl[2 * i]] - 1
        c = (a1 - a0) / (dist + 1)
        n = 1
        for j in range(bl[2 * i - 1] + 1, bl[2 * i]):
            s4[j] = a0 + c * n
            n += 1
    return s4, (s4 != 0).astype(np.float64)


SMOOTH_B = np.array([0.0078202080334971724, 0.015640416066994345, 0.0078202080334971724])
SMOOTH_A = np.array([1.0, -1.7347257688092754, 0.76600660094326412])


def smooth_f0(f0):
    """world/harvest.py:533-559: per voiced run, edge-held zero-phase 2nd-order Butterworth."""
    pad = 300
    buf = np.concatenate([np.zeros(pad), f0, np.zeros(pad)])
    bl = boundary_list(buf)
    out = buf.copy()
    for i in range(1, len(bl) // 2 + 1):
        st, ed = int(bl[2 * i - 2]), int(bl[2 * i - 1])
        ch = np.zeros(len(buf))
        ch[st : ed + 1] = buf[st : ed + 1]
        ch[:st] = ch[st]
        ch[ed + 1 :] = ch[ed]
        fwd = signal.lfilter(SMOOTH_B, SMOOTH_A, ch)
        bwd = signal.lfilter(SMOOTH_B, SMOOTH_A, fwd[::-1])[::-1]
        out[st : ed + 1] = bwd[st : ed + 1]
    return out[pad : len(out) - pad]


def harvest_np(x, fs, f0_floor=71, f0_ceil=800, frame_period=5, return_aux=False):
    """world/harvest.py:17-54 → dict(temporal_positions, f0, vuv)."""
    x = np.asarray(x, dtype=np.float64)
    n1 = C.frame_count(len(x), fs, 1)
    t1 = C.frame_times(n1, 1)
    y, fs_d = downsample_8k(x, fs)
    raw = raw_candidates(y, fs_d, fs, f0_floor, f0_ceil, t1)
    cands, most = detect_candidates(raw)
    cands = overlap_candidates(cands, most)
    cf0, csc = refine_candidates(y, fs_d, t1, cands, f0_floor, f0_ceil)
    pf0, psc = prune_isolated(cf0, csc)
    f0_1ms, vuv_1ms = fix_contour(pf0, psc)
    sm = smooth_f0(f0_1ms)
    nf = C.frame_count(len(x), fs, frame_period)
    tp = C.frame_times(nf, frame_period)
    pick = np.minimum(len(sm) - 1, C.half_up(tp * 1000)).astype(np.int64)
    res = {"temporal_positions": tp, "f0": sm[pick], "vuv": vuv_1ms[pick]}
    if return_aux:
        res["aux"] = {"y": y, "fs_d": fs_d, "raw": raw, "overlapped": cands, "refined_f0": cf0, "refined_score": csc,
                      "pruned_f0": pf0, "pruned_score": psc, "f0_1ms": f0_1ms, "smoothed_1ms": sm, "max_candidates": most}
    return res
