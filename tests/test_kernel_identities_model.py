"""CPU models of the three rewrites of round 4 that replaced a kernel's arithmetic by a cheaper equivalent, each checked
in NumPy against the form the reference computes (no GPU involved; the GPU suite checks the kernels end to end):

  * hv_refine_row (wh_harvest.hip): the two windowed spectra of GetRefinedF0 (world/harvest.py:169-211) at the harmonic
    bins as sums over sample PAIRS about the window centre.  What the refinement reads of them — |X|^2 and
    Re(X) Im(D) - Im(X) Re(D) — does not see the common unit factor that referring the phase to the centre introduces.
  * band_taps_fft_kernel / band_events_ols_kernel (wh_bands.h): a Harvest band filter (harvest.py:253-256) is symmetric
    about its centre tap, so rotated to put that tap at index 0 its spectrum is real; multiplying the tile spectrum by the
    real spectrum and reading the outputs half a filter length earlier is the reference's band-passed signal.
  * req_excite_kernel (wh_synthesis.hip): the periodic Requiem excitation (synthesisRequiem.py:51-61) gathered per output
    sample from the pulses that cover it, with the reference's clipped fancy-index semantics (only the last tap written
    to a clipped index survives), equals the scatter the reference performs — bit for bit, the sums run in pulse order
    either way — and the 64-ary wave search that finds a tile's first pulse is np.searchsorted(side='left')."""
import math

import numpy as np
import pytest

from oracle import pitch_harvest as H
from oracle import resynth as R


# ---- sums over sample pairs ------------------------------------------------------------------------------------------
def _window_pair(h, fs_d):
    """(w, dw) of half length h as the reference evaluates them for a frame whose sample picks step by one
    (harvest.py:176-187; the window argument is (j - h - 0.499)/fs_d whatever the frame time)."""
    ln = 2 * h + 1
    j = np.arange(ln)
    common = math.pi * ((j - h - 0.499) / fs_d) / (ln / fs_d)
    w = 0.42 + 0.5 * np.cos(2 * common) + 0.08 * np.cos(4 * common)
    prev = np.concatenate([[0.0], w[:-1]])
    nxt = np.concatenate([w[1:], [0.0]])
    dw = -((nxt - w) + (w - prev)) / 2
    dw[0] = -w[1] / 2
    dw[-1] = w[-2] / 2
    return w, dw


@pytest.mark.parametrize("f0c", [71.3, 118.0, 233.1, 640.5])
def test_pair_sums_about_the_window_centre_give_the_refinements_invariants(f0c):
    fs_d = 8000.0
    rng = np.random.default_rng(int(f0c))
    h = int(np.ceil(3 * fs_d / f0c / 2))
    ln = 2 * h + 1
    nfft = int(2 ** (np.ceil(np.log2(ln)) + 1))
    seg = rng.standard_normal(ln) + 0.3 * np.sin(2 * math.pi * f0c * np.arange(ln) / fs_d)
    w, dw = _window_pair(h, fs_d)
    assert np.max(np.abs(w - w[::-1])) > 1e-6  # the window is NOT symmetric: both halves are read
    bins = np.floor(f0c * nfft / fs_d * np.arange(1, 7) + 0.5).astype(int)
    bins = bins[bins < nfft // 2]
    # the reference: two zero-padded FFTs, read at the harmonic bins
    sp = np.fft.fft(seg * w, nfft)[bins]
    dsp = np.fft.fft(seg * dw, nfft)[bins]
    num_ref = sp.real * dsp.imag - sp.imag * dsp.real
    pow_ref = np.abs(sp) ** 2
    # the kernel: centre sample + pairs (h + m, h - m), twiddle index bin * m, phase referred to the centre
    a, d = seg * w, seg * dw
    m = np.arange(1, h + 1)
    ea, oa = a[h + m] + a[h - m], a[h + m] - a[h - m]
    ed, od = d[h + m] + d[h - m], d[h + m] - d[h - m]
    tw = np.exp(-2j * math.pi * np.outer(bins, m) / nfft)  # the table's entries (cos, -sin)
    xr = a[h] + (ea[None, :] * tw.real).sum(axis=1)
    xi = (oa[None, :] * tw.imag).sum(axis=1)
    dr = d[h] + (ed[None, :] * tw.real).sum(axis=1)
    di = (od[None, :] * tw.imag).sum(axis=1)
    num = xr * di - xi * dr
    power = xr * xr + xi * xi
    assert np.allclose(power, pow_ref, rtol=1e-11, atol=0)
    assert np.allclose(num, num_ref, rtol=1e-9, atol=1e-12 * np.max(pow_ref))
    # and with them the instantaneous frequencies the candidate is refined from
    inst_ref = (bins / nfft + num_ref / pow_ref / 2 / math.pi) * fs_d
    inst = (bins / nfft + num / power / 2 / math.pi) * fs_d
    assert np.max(np.abs(inst - inst_ref)) < 1e-8


def test_packed_refinement_geometry_round_trips():
    """refine_pack / refine_unpack (wh_harvest.hip): half length, first bin against a base that follows from the half
    length, harmonic count, the other bins as offsets from multiples of the first — 28 bits — for every candidate value a
    Harvest refinement can meet at the decimated rates in use."""
    for fs_d in (8000.0, 7350.0, 8820.0):
        f0 = np.concatenate([np.linspace(40.0, 1100.0, 20001), np.geomspace(40.0, 1100.0, 7001)])
        hwl = np.ceil(3 * fs_d / f0 / 2).astype(np.int64)
        nfft = (2 ** (np.ceil(np.log2(2 * hwl + 1)) + 1)).astype(np.int64)
        nh = np.minimum(np.floor(fs_d / 2 / f0), 6).astype(np.int64)
        bins = np.floor(f0[:, None] * nfft[:, None] / fs_d * np.arange(1, 7)[None, :] + 0.5).astype(np.int64)
        base = (np.float32(1.5) * nfft.astype(np.float32) / hwl.astype(np.float32)).astype(np.int64) - 1
        b0 = bins[:, 0] - base
        off = np.array([0, 1, 3, 3, 3, 3])
        d = bins - np.arange(1, 7)[None, :] * bins[:, :1] + off[None, :]
        lim = np.array([1, 4, 8, 8, 8, 8])
        fits = (b0 >= 0) & (b0 < 4) & (hwl < 512) & np.all((d >= 0) & (d < lim[None, :]), axis=1)
        assert fits.mean() > 0.999, fs_d  # (an item that does not fit is a class of its own: slower, not wrong)
        back = np.arange(1, 7)[None, :] * (base + b0)[:, None] + d - off[None, :]
        assert np.array_equal(back[fits], bins[fits])
        assert np.all(nh[fits] < 8)


# ---- real tap spectra ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bf", [65.0, 131.7, 402.3, 890.1])
def test_zero_phase_tap_spectrum_is_real_and_filters_like_the_reference(bf):
    fs_d = 8000.0
    n = 4096
    taps, h = H.band_pass_taps(bf, fs_d)
    assert len(taps) == 2 * h + 1
    rot = np.zeros(n)
    rot[: h + 1] = taps[h:]
    rot[n - h:] = taps[:h]
    spec = np.fft.rfft(rot)
    assert np.max(np.abs(spec.imag)) <= 1e-13 * np.max(np.abs(spec.real))  # the taps' rounding asymmetry
    rng = np.random.default_rng(int(bf))
    hmax = 246                      # longest filter of the default band set: the block starts hmax samples early
    y = rng.standard_normal(3 * n)
    t0 = 5000                       # first sample of the tile
    valid = 3584 + 2
    block = y[t0 - hmax: t0 - hmax + n]
    out = np.fft.irfft(np.fft.rfft(block) * spec.real, n)
    got = out[hmax + 1: hmax + 1 + valid]             # the walker's read position: H + 1
    # the reference: convolution with the taps as they lie, delayed by h + 1 (harvest.py:257-259)
    full = np.convolve(y, taps)
    ref = full[(h + 1) + t0 + np.arange(valid)]
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))


# ---- gathered Requiem excitation ---------------------------------------------------------------------------------------
def _first_pulse_at(pi, lo):
    """The 64-ary search of first_pulse_at (wh_synthesis.hip), lane by lane."""
    base, n = 0, len(pi)
    while n > 0:
        stride = (n + 63) // 64
        c = 0
        for lane in range(64):
            idx = base + lane * stride
            if idx < base + n and pi[idx] < lo:
                c += 1
        if stride == 1:
            base += c
            break
        if c == 0:
            break
        nb = base + (c - 1) * stride + 1
        left = base + n - nb
        n = min(stride - 1, left)
        base = nb
    return base


def test_wave_search_is_searchsorted_left():
    rng = np.random.default_rng(5)
    for count in list(range(0, 70)) + [127, 128, 129, 1000, 4095, 4096, 4097, 20064]:
        pi = np.sort(rng.integers(1, 160000, size=count))
        for lo in np.concatenate([[-500, 0, 1, 160001], rng.integers(-300, 160300, size=12), pi[:3], pi[-3:] + 1]):
            assert _first_pulse_at(pi, lo) == np.searchsorted(pi, lo, side="left"), (count, lo)


def _gathered_periodic(ny, pidx, gain, wm, seed):
    """req_excite_kernel's periodic part, sample by sample."""
    pfft, nb = seed.shape
    out = np.zeros(ny)
    for i in range(ny):
        tgt = i + 1
        acc = 0.0
        if tgt < ny:
            k = _first_pulse_at(pidx, tgt - pfft // 2 + 1 - 1)  # (any start at or before the first covering pulse)
            while k < len(pidx) and pidx[k] <= tgt + pfft // 2 - 1:
                mm = tgt - pidx[k] + pfft // 2 - 1
                if gain[k] != 0.0 and 0 <= mm < pfft:
                    r = 0.0
                    for b in range(nb):
                        r += seed[mm, b] * wm[k, b]
                    acc += r * gain[k]
                k += 1
        else:
            for k in range(_first_pulse_at(pidx, ny - pfft // 2), len(pidx)):
                if gain[k] != 0.0:
                    r = 0.0
                    for b in range(nb):
                        r += seed[pfft - 1, b] * wm[k, b]
                    acc += r * gain[k]
        out[i] = acc
    return out


@pytest.mark.parametrize("case", range(6))
def test_gathered_periodic_excitation_equals_the_references_scatter(case):
    rng = np.random.default_rng(100 + case)
    pfft, nb = 64, 3
    ny = int(rng.integers(90, 400))
    seed = rng.standard_normal((pfft, nb))
    count = int(rng.integers(3, 40))
    # pulses anywhere in [1, ny], including within half a seed length of both ends (the clipped cases)
    pidx = np.sort(rng.choice(np.arange(1, ny + 1), size=min(count, ny), replace=False))
    if case % 2 == 0:
        pidx[0], pidx[-1] = 1, ny
    ap = rng.uniform(0.0, 1.0, size=(len(pidx), nb))
    skip = rng.uniform(size=len(pidx)) < 0.25  # unvoiced at the pulse, or lowest band above 0.999
    nxt = np.concatenate([pidx[1:], pidx[-1:]])
    gain = np.where(skip, 0.0, np.sqrt(np.maximum(1, nxt - pidx)))
    wm = 1 - ap
    # the reference's scatter: pulses in order, clipped fancy-index +=
    base_index = np.arange(-pfft // 2 + 1, pfft // 2 + 1)
    ref = np.zeros(ny)
    for k in range(len(pidx)):
        if skip[k]:
            continue
        R._ola(ref, int(pidx[k]), base_index, R._seed_mix(seed, ap[k]) * np.sqrt(max(1, int(nxt[k] - pidx[k]))))
    got = _gathered_periodic(ny, pidx, gain, wm, seed)
    assert np.array_equal(got, ref)


# ---- round 5: overlap-add without atomics (response_kernel rows + response_gather_kernel; wh_synthesis.hip RunState) --------
def _row_offsets(ny, pidx, n, run, capacity):
    """pulse_rows_kernel: rows one behind the other, row r of 1 + (end_r - start_r) doubles at the exclusive prefix of
    the lengths; -1 (dropped) where the region is full."""
    count = len(pidx)
    offs, at = [], 0
    for r in range((count + run - 1) // run):
        kf, kl = r * run, min((r + 1) * run, count) - 1
        start, end = max(int(pidx[kf]) - n // 2 + 1, 1), min(int(pidx[kl]) + n // 2 + 1, ny)
        ln = 1 + max(end - start, 0)
        offs.append(at if at + ln <= capacity else -1)
        at += ln
    return offs, at


def _rows_of_runs(ny, pidx, resp, n, run, offs, capacity):
    """response_kernel's overlap-add, thread loops flattened: runs of `run` consecutive pulses accumulate in an n-sample
    ring; what leaves the ring's window goes to the run's row (slot 0: the last sample's share, slot 1 + (t - start_r):
    sample t).  Unwritten slots stay NaN so that a gather that reads one is caught."""
    count = len(pidx)
    rows = np.full(capacity, np.nan)
    for r in range((count + run - 1) // run):
        if offs[r] < 0:
            continue
        ring = np.zeros(n)
        any_, win_start, row_start, last = False, 0, 1, 0.0
        base = offs[r]
        for k in range(r * run, min((r + 1) * run, count)):
            s1 = int(pidx[k]) - n // 2 + 1
            if any_:
                e = min(s1, win_start + n)
                for tgt in range(max(win_start, 1), min(e, ny)):          # ring_flush
                    rows[base + 1 + (tgt - row_start)] = ring[tgt & (n - 1)]
                    ring[tgt & (n - 1)] = 0.0
                for tgt in range(win_start + n, min(s1, ny)):              # pulses more than n apart
                    rows[base + 1 + (tgt - row_start)] = 0.0
            else:
                row_start = max(s1, 1)
            any_, win_start = True, s1
            for mm in range(n):
                tgt = s1 + mm
                if tgt < 1:
                    continue
                if tgt < ny:
                    ring[tgt & (n - 1)] += resp[k][mm]
                elif mm == n - 1:
                    last += resp[k][mm]
        if any_:
            for tgt in range(max(win_start, 1), min(win_start + n, ny)):
                rows[base + 1 + (tgt - row_start)] = ring[tgt & (n - 1)]
            rows[base] = last
    return rows


def _gather_rows(ny, pidx, rows, n, run, offs, tile=256):
    """response_gather_kernel, one tile of samples at a time."""
    count = len(pidx)
    n_runs = (count + run - 1) // run
    y = np.zeros(ny)
    for n0 in range(0, ny, tile):
        k0 = _first_pulse_at(pidx, n0 + 1 - n // 2)
        k_end = _first_pulse_at(pidx, ny - n // 2)
        for i in range(n0, min(n0 + tile, ny)):
            tgt, acc = i + 1, 0.0
            if tgt < ny:
                for r in range(k0 // run, n_runs):
                    kf = r * run
                    kl = min(kf + run, count) - 1
                    s1f = int(pidx[kf]) - n // 2 + 1
                    if s1f > n0 + tile:
                        break
                    start, end = max(s1f, 1), min(int(pidx[kl]) + n // 2 + 1, ny)
                    if offs[r] >= 0 and start <= tgt < end:
                        acc += rows[offs[r] + 1 + (tgt - start)]
            else:
                for r in range(k_end // run, n_runs):
                    if offs[r] >= 0:
                        acc += rows[offs[r]]
            y[i] = acc
    return y


@pytest.mark.parametrize("case", range(8))
def test_response_rows_and_gather_equal_the_references_scatter(case):
    """Integer-valued responses: every association of the sum is exact, so the gathered rows must EQUAL the reference's
    pulse-by-pulse clipped fancy-index += (world/synthesis.py:67-81), including both clipped ends, pulses closer than,
    as far as and farther apart than the transform length, and a pulse count that is no multiple of the run."""
    rng = np.random.default_rng(300 + case)
    n, run = 64, (3 if case % 2 else 6)
    ny = int(rng.integers(300, 900))
    count = int(rng.integers(1, 40))
    if case == 5:  # sparse: gaps longer than n between pulses
        pidx = np.sort(rng.choice(np.arange(1, ny + 1, 90), size=min(count, len(np.arange(1, ny + 1, 90))), replace=False))
    else:
        pidx = np.sort(rng.choice(np.arange(1, ny + 1), size=min(count, ny), replace=False))
    if case % 3 == 0:
        pidx[0], pidx[-1] = 1, ny
    resp = rng.integers(-9, 10, size=(len(pidx), n)).astype(np.float64)
    base_index = np.arange(-n // 2 + 1, n // 2 + 1)
    ref = np.zeros(ny)
    for k in range(len(pidx)):
        R._ola(ref, int(pidx[k]), base_index, resp[k])
    offs, need = _row_offsets(ny, pidx, n, run, 10 ** 9)
    assert min(offs) >= 0 and len(set(offs)) == len(offs)
    rows = _rows_of_runs(ny, pidx, resp, n, run, offs, need)   # a region of exactly the size the rows take
    assert not np.any(np.isnan(rows))                          # every slot of every row is written, once
    for tile in (256, 1024, 96):
        got = _gather_rows(ny, pidx, rows, n, run, offs, tile)
        assert not np.any(np.isnan(got))
        assert np.array_equal(got, ref)
    if len(offs) > 2:  # a region that is too small: the rows that fit are intact, the others dropped (and flagged)
        short, _ = _row_offsets(ny, pidx, n, run, need - 1)
        assert short[:-1] == offs[:-1] and short[-1] == -1


def _req_rows(ny, nf, hop, n, runf, resp):
    """req_filter_kernel<N, RUNF>'s rows: frames 2 .. nf-2 in runs of runf, each run's responses summed in frame order
    over (runf - 1) hop + n samples; slot 0 = the run's share of the last sample.  Unwritten slots stay NaN."""
    frames = max(nf - 3, 0)
    n_runs = (frames + runf - 1) // runf
    w = (runf - 1) * hop + n + 1
    rows = np.full(n_runs * w, np.nan)
    for r in range(n_runs):
        i0 = r * runf + 2
        i1 = min(i0 + runf - 1, nf - 2)
        a_r = (i0 - 2) * hop + 1
        acc, last = np.zeros(w - 1), 0.0
        for i in range(i0, i1 + 1):  # the run's sums: an accumulator over the run's span, written to the row once
            origin = (i - 1) * hop - (hop - 1)
            for mm in range(n):
                acc[origin - a_r + mm] += resp[i][mm] if origin + mm < ny else 0.0
            if origin + n - 1 >= ny:
                last += resp[i][n - 1]
        rows[r * w + 1:(r + 1) * w] = acc
        rows[r * w] = last
    return rows, n_runs, w


def _req_gather(ny, nf, hop, n, runf, rows, n_runs, w):
    adv = runf * hop
    y = np.zeros(ny)
    for i in range(ny):
        tgt, acc = i + 1, 0.0
        if tgt < ny:
            r_hi = min((tgt - 1) // adv, n_runs - 1)
            x = tgt - (w - 1)
            r_lo = 0 if x <= 0 else (x - 1) // adv + 1
            for r in range(r_lo, r_hi + 1):
                acc += rows[r * w + 1 + (tgt - (r * adv + 1))]
        else:
            r_lo = max(int((ny - n) / adv) - 1, 0)  # (C++ division truncates towards zero)
            for r in range(r_lo, n_runs):
                acc += rows[r * w]
        y[i] = acc
    return y


@pytest.mark.parametrize("case", range(8))
def test_requiem_frame_rows_and_gather_equal_the_references_overlap_add(case):
    """req_filter_kernel / req_gather_kernel against the reference's frame loop (world/synthesisRequiem.py:83-100:
    y[clip(origin + arange(fft))] += response, frames 2 .. F-2), integer-valued responses so that every association of
    the sum is exact: runs of 8, 3 and 1 frames, output lengths that cut the last responses short, fewer than 4 frames."""
    rng = np.random.default_rng(400 + case)
    n = 64
    hop = int(rng.integers(3, 12))
    runf = (8, 3, 1, 8)[case % 4]
    nf = int(rng.integers(2, 40)) if case != 6 else 3
    ny = max(2, (nf - 1) * hop + int(rng.integers(-hop, hop + 1)) + 1)  # about one hop around the last frame time
    resp = rng.integers(-9, 10, size=(nf + 1, n)).astype(np.float64)
    ref = np.zeros(ny)
    for i in range(2, nf - 1):  # i = 2 .. nf-2
        origin = (i - 1) * hop - (hop - 1)
        R._ola(ref, origin, np.arange(n), resp[i])
    rows, n_runs, w = _req_rows(ny, nf, hop, n, runf, resp)
    got = _req_gather(ny, nf, hop, n, runf, rows, n_runs, w)
    assert not np.any(np.isnan(got)) and np.array_equal(got, ref)
