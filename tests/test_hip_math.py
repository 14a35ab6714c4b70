"""GPU: the spectral kernels' own log / exp / sincospi (csrc/wh_math.h, through wh_math_probe) against NumPy: a few ulp over the
ranges the kernels feed them and far beyond, special values passed through."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _probe(which, x):
    from world import _hip

    rt = _hip.Runtime.get()
    x_d = rt.to_device(np.ascontiguousarray(x, dtype=np.float64))
    out = rt.empty((len(x) * (2 if which == 2 else 1),))
    _hip.check(rt.lib.wh_math_probe(rt.ctx, rt.stream(), which, rt.ptr(x_d), rt.ptr(out), len(x)))
    return out.cpu().numpy()


def test_log():
    rng = np.random.RandomState(1)
    x = np.concatenate([10 ** rng.uniform(-300, 300, 400000), 1 + rng.uniform(-1e-3, 1e-3, 100000), rng.uniform(0.5, 2.0, 100000),
                        [1.0, 2.0, 0.5, np.e, 2.2250738585072014e-308, 5e-324, 1.7976931348623157e308, 0.7071067811865476, 1.4142135623730951]])
    got, ref = _probe(0, x), np.log(x)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    assert np.max(err[ref != 0]) < 5e-16
    assert got[x == 1.0][0] == 0.0
    sp = _probe(0, np.array([0.0, -1.0, np.inf, np.nan]))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3])


def test_exp():
    rng = np.random.RandomState(2)
    x = np.concatenate([rng.uniform(-700, 700, 400000), rng.uniform(-1, 1, 100000), rng.uniform(-1e-8, 1e-8, 1000), [0.0, 1.0, -1.0, 709.0, -745.0]])
    got, ref = _probe(1, x), np.exp(x)
    ok = ref > 1e-300  # (denormal results round differently; nothing in the kernels gets near)
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 5e-16
    sp = _probe(1, np.array([800.0, -800.0, np.inf, -np.inf, np.nan, 1e12, -1e12]))
    assert sp[0] == np.inf and sp[1] == 0.0 and sp[2] == np.inf and sp[3] == 0.0 and np.isnan(sp[4]) and sp[5] == np.inf and sp[6] == 0.0


def test_sincospi():
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(-4, 4, 400000), rng.uniform(-1e6, 1e6, 100000), np.arange(-8, 9) * 0.25, rng.uniform(-1e-9, 1e-9, 1000)])
    got = _probe(2, x).reshape(-1, 2)
    # reference in extended precision on the reduced argument (np.sin(pi * x) loses the digits of large x)
    r = (x - 2 * np.round(0.5 * x)).astype(np.longdouble)
    ref_s, ref_c = np.sin(np.pi * r, dtype=np.longdouble), np.cos(np.pi * r, dtype=np.longdouble)
    pi_l = np.longdouble("3.14159265358979323846264338327950288")
    ref_s, ref_c = np.sin(pi_l * r), np.cos(pi_l * r)
    assert np.max(np.abs(got[:, 0] - ref_s.astype(np.float64))) < 3e-16
    assert np.max(np.abs(got[:, 1] - ref_c.astype(np.float64))) < 3e-16
    exact = _probe(2, np.array([0.0, 0.5, 1.0, 1.5, 2.0, -0.5])).reshape(-1, 2)
    assert np.array_equal(exact, np.array([[0.0, 1.0], [1.0, 0.0], [0.0, -1.0], [-1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]]) + 0.0) or np.max(np.abs(exact - np.array([[0, 1], [1, 0], [0, -1], [-1, 0], [0, 1], [-1, 0]]))) == 0.0
    sp = _probe(2, np.array([np.nan, np.inf])).reshape(-1, 2)
    assert np.isnan(sp).all()
