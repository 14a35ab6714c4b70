"""CPU: a NumPy model of the tile-parallel exact phase scan (python-world_amd/csrc/wh_synthesis.hip: xs_*_kernel) —
the argument that makes np.cumsum parallel, checked bit for bit without a GPU.

Sequential semantics: a[j+1] = fl(a[j] + x[j]), x >= 0.  While a stays in one binade [2^k, 2^(k+1)) every partial sum is
a multiple of q = 2^(k-52) and fl(a + x) = a + q*RN(x/q), the rounding of the addend being independent of a unless x/q is
an exact tie.  A tile that lies inside one binade and holds no tie therefore needs only its carry, as an additive
constant; the model below follows the five device passes, including the sequential walk that verifies every tile with
the EXACT carry before accepting it."""
import numpy as np

TOP = 2.0 ** 53


def _seq(x, carry):
    """the sequential-equivalent fallback (device: exact_cumsum_block): plain left-to-right float adds"""
    out = np.empty(len(x))
    a = carry
    for j, v in enumerate(x):
        a = a + v
        out[j] = a
    return out, a


def tile_scan_model(x, tile):
    n = len(x)
    nt = (n + tile - 1) // tile
    out = np.empty(n)
    # passes 1-2: plain floating-point tile sums and their running sums (any order: only the binade is taken from them)
    s = np.array([np.sum(x[t * tile:(t + 1) * tile]) for t in range(nt)])
    approx = np.concatenate([[0.0], np.cumsum(s)[:-1]])
    # pass 3: per tile, with the binade of the approximate carry: integer total, flags
    T = np.zeros(nt)
    sh = np.zeros(nt, dtype=np.int64)
    flag = np.ones(nt, dtype=bool)
    for t in range(nt):
        a = approx[t]
        if not (a > 0 and np.isfinite(a)) or a < 2.0 ** -1022:
            continue
        e = int(np.floor(np.log2(a)))
        if not (2.0 ** e <= a < 2.0 ** (e + 1)):   # log2 rounding at a power of two
            e = e + 1 if a >= 2.0 ** (e + 1) else e - 1
        sh[t] = 52 - e
        sc = np.minimum(np.ldexp(x[t * tile:(t + 1) * tile], int(sh[t])), 2.0 ** 54)
        fl = np.floor(sc)
        fr = sc - fl
        bad = np.any(fr == 0.5) or np.any(sc >= 2.0 ** 52) or np.any(x[t * tile:(t + 1) * tile] < 0)
        T[t] = np.sum(fl + (fr > 0.5))
        flag[t] = bad or not (T[t] < 2.0 ** 52)
    # pass 4: the sequential walk with exact carries; pass 5 (closed form) applied as tiles are accepted
    a = 0.0
    irregular = 0
    for t in range(nt):
        seg = x[t * tile:(t + 1) * tile]
        ok = not flag[t]
        if ok:
            ok = a >= 2.0 ** -1022 and 2.0 ** (52 - sh[t]) <= a < 2.0 ** (53 - sh[t])   # the true carry is in the assumed binade
        if ok:
            v = np.ldexp(a, int(sh[t])) + T[t]                                             # integers below 2^53: exact
            ok = v < TOP
        if ok:
            sc = np.ldexp(seg, int(sh[t]))
            fl = np.floor(sc)
            r = fl + (sc - fl > 0.5)
            out[t * tile:t * tile + len(seg)] = np.ldexp(np.ldexp(a, int(sh[t])) + np.cumsum(r), -int(sh[t]))
            a = float(np.ldexp(v, -int(sh[t])))
        else:
            irregular += 1
            out[t * tile:t * tile + len(seg)], a = _seq(seg, a)
    return out, irregular


def _cases():
    rng = np.random.RandomState(11)
    t = np.arange(60001)
    yield 2 * np.pi * (120 + 40 * np.sin(t / 9000.0)) / 16000.0, 512      # phase increments: ~17 binades, a few ties
    yield np.full(30001, 2 * np.pi * 500 / 16000.0), 512                   # the unvoiced default: constant increment
    yield rng.uniform(0.0, 1.0, 20000), 256
    yield 10.0 ** rng.uniform(-12, 3, 8000), 128                           # 15 decades: crossings everywhere
    yield np.concatenate([np.zeros(37), rng.uniform(0, 1e-3, 3000)]), 128  # leading zeros
    x = np.full(9000, 0.75)
    x[1::2] = 2.0 ** -45 * 3                                               # exact ties, again and again
    yield x, 256
    y = np.ones(3000)
    y[::3] = 2.0 ** -42
    y[1::3] = 2.0 ** -43
    yield y, 128


def test_tile_scan_model_equals_cumsum_bitwise():
    for x, tile in _cases():
        got, _ = tile_scan_model(x, tile)
        want = np.cumsum(x)
        assert np.array_equal(got.view(np.int64), want.view(np.int64))


def test_smooth_increments_leave_few_irregular_tiles():
    """the point of the scheme: for phase-increment data almost every tile takes the closed form"""
    t = np.arange(400001)
    x = 2 * np.pi * (150 + 50 * np.sin(t / 7000.0)) / 48000.0
    got, irregular = tile_scan_model(x, 4096)
    assert np.array_equal(got.view(np.int64), np.cumsum(x).view(np.int64))
    assert irregular <= 30 and irregular < (len(x) + 4095) // 4096 // 2
