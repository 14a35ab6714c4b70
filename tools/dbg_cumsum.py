import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
from world import _hip
rt = _hip.Runtime.get()
def run(seg):
    off = np.array([0, len(seg)], dtype=np.int64)
    d = rt.to_device(seg)
    _hip.check(rt.lib.wh_cumsum_exact(rt.ctx, rt.stream(), rt.ptr(d), off.ctypes.data_as(ctypes.c_void_p), 1))
    return d.cpu().numpy()
rng = np.random.RandomState(1234)
cases = {}
t = np.arange(400001)
cases["phase"] = 2 * np.pi * (120 + 40 * np.sin(t / 9000.0)) / 16000.0
cases["const"] = np.full(160001, 2 * np.pi * 500 / 16000.0)
cases["uni"] = rng.uniform(0.0, 1.0, 70000)
cases["decades"] = 10.0 ** rng.uniform(-12, 3, 50000)
cases["zeros"] = np.concatenate([np.zeros(37), rng.uniform(0, 1e-3, 5000)])
x = np.full(30000, 0.75); x[1::2] = 2.0 ** -45 * 3
cases["ties1"] = x
y = np.ones(5000); y[::3] = 2.0 ** -42; y[1::3] = 2.0 ** -43
cases["ties2"] = y
for n in (1, 2, 31, 33, 2047, 2048, 2049, 4097):
    cases["n%d" % n] = rng.uniform(0.01, 0.1, n)
for name, seg in cases.items():
    got = run(seg); want = np.cumsum(seg)
    bad = np.nonzero(got.view(np.int64) != want.view(np.int64))[0]
    if len(bad):
        i = bad[0]
        print(name, "MISMATCH first at", i, "of", len(seg), "count", len(bad), "got", got[i].hex(), "want", want[i].hex(),
              "prev", want[i - 1].hex() if i else None, "x", seg[i].hex())
    else:
        print(name, "ok")
