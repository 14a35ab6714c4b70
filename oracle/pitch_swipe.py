"""ORACLE (test infrastructure) — NumPy restatement of SWIPE' as the reference implements it
(world/swipe.py:9-169; dispatched from world/main.py:45-46,134-135 for f0_method='swipe').
Vectorised over frames and candidates; used only by tests (pinned by tests/golden/golden_swipe.npz)."""
from decimal import ROUND_HALF_UP, Decimal

import numpy as np
from scipy import interpolate


def _round_half_up(v):
    return int(Decimal(float(v)).quantize(0, ROUND_HALF_UP))  # swipe.py:107-114


def hz2erbs(hz):
    return 21.4 * np.log10(1 + hz / 229)  # swipe.py:150-152


def erbs2hz(erbs):
    return (10 ** (erbs / 21.4) - 1) * 229  # swipe.py:153-155


def sieve_as_reference(n):
    """swipe.py:158-172 (`sieve`).  NOT the primes in [2, n]: the reference strikes multiples of p only `while p <
    sqrt(n)`, so for n = p*p (p prime) the number n itself survives — sieve(9) = [2, 3, 5, 7, 9], sieve(25) ends in 25.
    Restated as: the numbers in [2, n] with no divisor d, 2 <= d, d*d <= m, except that for m == n only d*d < n counts."""
    out = []
    for m in range(2, n + 1):
        if all(m % d for d in range(2, m) if (d * d < m or (d * d == m and m != n))):
            out.append(m)
    return out


def candidate_kernel(f, pc):
    """swipe.py:127-149 without the final dot product: the K+-normalised kernel of one pitch candidate."""
    n = int(np.fix(f[-1] / pc - 0.75))
    k = np.zeros(len(f))
    q = f / pc
    for i in [1] + sieve_as_reference(n):
        a = np.abs(q - i)
        peak = a < 0.25
        k[peak] = np.cos(2 * np.pi * q[peak])
        valley = np.logical_and(0.25 < a, a < 0.75)
        k[valley] = k[valley] + np.cos(2 * np.pi * q[valley]) / 2
    k *= np.sqrt(1 / f)
    k /= np.linalg.norm(k[k > 0])
    return k


def stft_complex(xzp, nfft, window, hop):
    """matplotlib.mlab.specgram(mode='complex') framing: segments every `hop` samples, no detrending, scaled by
    1/sum(window); times of the segment centres."""
    n_seg = (len(xzp) - (nfft - hop)) // hop
    idx = np.arange(nfft)[:, None] + hop * np.arange(n_seg)[None, :]
    spec = np.fft.rfft(xzp[idx] * window[:, None], axis=0) / window.sum()
    return spec, None, (np.arange(n_seg) * hop + nfft / 2)


def swipe_np(fs, x, plim=(71, 800), dt=0.005, sTHR=float("-inf")):
    plim = np.array(plim, dtype=np.float64)
    num = int(1000 * len(x) / fs / (dt * 1000) + 1)
    t = np.arange(0, num) * dt
    log2pc = np.arange(np.log2(plim[0]) * 96, np.log2(plim[-1]) * 96) * (1 / 96)
    pc = 2 ** log2pc
    S = np.zeros((len(pc), len(t)))
    log_ws = [_round_half_up(e) for e in np.log2(4 * 2 * fs / plim)]
    ws = 2 ** np.arange(log_ws[0], log_ws[1] - 1, -1)
    p0 = 4 * 2 * fs / ws
    d = 1 + log2pc - np.log2(4 * 2 * fs / ws[0])
    f_erbs = erbs2hz(np.arange(hz2erbs(pc[0] / 4), hz2erbs(fs / 2), 0.1))
    for i, w_size in enumerate(ws):
        w_size = int(w_size)
        dn = _round_half_up(4 * fs / p0[i])
        xzp = np.r_[np.zeros(w_size // 2), x, np.zeros(int(dn + w_size / 2))]
        window = np.hanning(w_size + 2)[1:-1]
        overlap = max(0, np.round(w_size - dn))
        X, _, ti = stft_complex(xzp, w_size, window, int(w_size - overlap))
        ti = np.r_[0, (ti / fs)[:-1]]
        f = np.arange(w_size // 2 + 1) * fs / w_size
        M = np.maximum(0, interpolate.interp1d(f, np.abs(X.T), kind="cubic")(f_erbs)).T
        L = np.sqrt(M)
        if i == len(ws) - 1:
            j = np.where(d - (i + 1) > -1)[0]
            k = np.where(d[j] - (i + 1) < 0)[0]
        elif i == 0:
            j = np.where(d - (i + 1) < 1)[0]
            k = np.where(d[j] - (i + 1) > 0)[0]
        else:
            j = np.where(np.abs(d - (i + 1)) < 1)[0]
            k = np.arange(len(j))
        den = np.sqrt(np.sum(L * L, axis=0))
        L = L / np.where(den == 0, 2.220446049250313e-16, den)
        Si = np.stack([candidate_kernel(f_erbs, pc[c]) @ L for c in j])
        if Si.shape[1] > 1:
            Si = interpolate.interp1d(ti, Si, bounds_error=False, fill_value=np.nan)(t)
        else:
            Si = np.full((len(Si), len(t)), np.nan)
        mu = np.ones(len(j))
        mu[k] = 1 - np.abs(d[j[k]] - i - 1)
        S[j, :] += mu[:, None] * Si
    p = np.full(S.shape[1], np.nan)
    for col in range(S.shape[1]):
        s = np.max(S[:, col])
        i = int(np.argmax(S[:, col]))
        if s < sTHR:
            continue
        if i == 0 or i == len(pc) - 1:
            p[col] = pc[0]
            continue
        I = np.arange(i - 1, i + 2)
        tc = 1 / pc[I]
        ntc = (tc / tc[1] - 1) * 2 * np.pi
        c = np.polyfit(ntc, S[I, col], 2)
        ftc = 1 / (2 ** np.arange(np.log2(pc[I[0]]), np.log2(pc[I[2]]) + 1 / 12 / 64, 1 / 12 / 64))
        nftc = (ftc / tc[1] - 1) * 2 * np.pi
        kbest = int(np.argmax(np.polyval(c, nftc)))
        p[col] = 2 ** (np.log2(pc[I[0]]) + kbest / 12 / 64)
    p[np.isnan(p)] = 0
    return {"temporal_positions": t, "f0": p, "vuv": (p > 0).astype(np.float64)}
