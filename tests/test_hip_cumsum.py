"""GPU: the phase accumulator's scan (wh_cumsum_exact) equals np.cumsum BIT FOR BIT — it is a parallel integer
prefix sum per binade of the running sum, so the cases that matter are binade crossings, exact rounding ties,
leading zeros and segment lengths around the tile size."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_segments(segs):
    from world import _hip

    rt = _hip.Runtime.get()
    off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    d = rt.to_device(np.concatenate(segs) if len(segs) else np.zeros(0))
    _hip.check(rt.lib.wh_cumsum_exact(rt.ctx, rt.stream(), rt.ptr(d), off.ctypes.data_as(ctypes.c_void_p), len(segs)))
    out = d.cpu().numpy()
    return [out[off[i]:off[i + 1]] for i in range(len(segs))]


def test_matches_numpy_cumsum_bitwise():
    rng = np.random.RandomState(1234)
    segs = []
    # phase-increment-like data: 2*pi*f/fs with slowly varying f, long enough for ~17 binades and a dozen ties
    t = np.arange(400001)
    segs.append(2 * np.pi * (120 + 40 * np.sin(t / 9000.0)) / 16000.0)
    segs.append(np.full(160001, 2 * np.pi * 500 / 16000.0))          # the unvoiced default: a constant increment
    segs.append(rng.uniform(0.0, 1.0, 70000))                         # arbitrary magnitudes
    segs.append(10.0 ** rng.uniform(-12, 3, 50000))                   # 15 decades: crossings everywhere
    segs.append(np.concatenate([np.zeros(37), rng.uniform(0, 1e-3, 5000)]))  # leading zeros
    for n in (0, 1, 2, 31, 32, 33, 2047, 2048, 2049, 4096, 4097):     # around the tile / lane-run sizes
        segs.append(rng.uniform(0.01, 0.1, n))
    # exact ties: increments that are odd multiples of half an ulp of the running sum
    x = np.full(30000, 0.75)
    x[1::2] = 2.0 ** -45 * 3          # sum ~ 1e4 -> ulp 2^-39: these land on .5-ulp boundaries again and again
    segs.append(x)
    y = np.ones(5000)
    y[::3] = 2.0 ** -42               # ulp of sums in [2^10, 2^11) is 2^-42: half-ulp ties need 2^-43 ...
    y[1::3] = 2.0 ** -43
    segs.append(y)
    got = run_segments(segs)
    for g, s in zip(got, segs):
        want = np.cumsum(s)
        assert g.shape == want.shape
        assert np.array_equal(g.view(np.int64), want.view(np.int64))


def test_many_segments_ragged():
    rng = np.random.RandomState(7)
    segs = [rng.uniform(0.02, 0.3, rng.randint(1, 9000)) for _ in range(40)]
    for g, s in zip(run_segments(segs), segs):
        assert np.array_equal(g.view(np.int64), np.cumsum(s).view(np.int64))
