"""The C-ABI is usable by a host that has no PyTorch: a subprocess that never imports torch drives wh_cheaptrick
with nothing but wh_ctx_create / wh_malloc / wh_memcpy_* / wh_memset / wh_stream_sync (include/world_hip.h:37-47)
and reproduces the reference fixture."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms

DRIVER = r'''
import ctypes, sys
import numpy as np
lib_path, golden, out_path = sys.argv[1:4]
lib = ctypes.CDLL(lib_path)
vp, dbl, i64 = ctypes.c_void_p, ctypes.c_double, ctypes.c_int64
lib.wh_last_error.restype = ctypes.c_char_p
def ok(rc):
    if rc != 0:
        raise RuntimeError(lib.wh_last_error().decode())
lib.wh_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
lib.wh_free.argtypes = [vp]
lib.wh_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t, vp]
lib.wh_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t, vp]
lib.wh_memset.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
lib.wh_stream_sync.argtypes = [vp]
lib.wh_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
lib.wh_ctx_destroy.argtypes = [vp]
lib.wh_batch_create.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.POINTER(vp)]
lib.wh_batch_destroy.argtypes = [vp]
lib.wh_num_frames.restype = i64
lib.wh_num_frames.argtypes = [i64, dbl, dbl]
lib.wh_cheaptrick.argtypes = [vp, vp, vp, vp, vp, vp, vp, dbl, ctypes.c_int, dbl, vp, vp]
lib.wh_take_flags.argtypes = [vp, vp, vp]
n = ctypes.c_int(0)
ok(lib.wh_device_count(ctypes.byref(n)))
assert n.value >= 1
g = np.load(golden)
x = np.ascontiguousarray(g["x"], dtype=np.float64)
fs = float(g["fs"])
tp = np.ascontiguousarray(g["tp"]); f0 = np.ascontiguousarray(g["stonemask_f0"]); vuv = np.ascontiguousarray(g["dio_vuv"])
nf = len(tp)
assert lib.wh_num_frames(len(x), fs, 5.0) == nf
ctx = vp()
ok(lib.wh_ctx_create(0, ctypes.byref(ctx)))
def dev(a):
    p = vp()
    ok(lib.wh_malloc(ctypes.byref(p), a.nbytes))
    ok(lib.wh_memcpy_h2d(p, a.ctypes.data_as(vp), a.nbytes, None))
    return p
x_d, tp_d, f0_d, vuv_d = dev(x), dev(tp), dev(f0), dev(vuv)
fft = 1024
k = fft // 2 + 1
spec_d = vp()
ok(lib.wh_malloc(ctypes.byref(spec_d), nf * k * 8))
ok(lib.wh_memset(spec_d, 0, nf * k * 8, None))
x_off = np.array([0, len(x)], dtype=np.int64); f_off = np.array([0, nf], dtype=np.int64)
b = vp()
ok(lib.wh_batch_create(ctx, 1, x_off.ctypes.data_as(vp), f_off.ctypes.data_as(vp), ctypes.byref(b)))
ok(lib.wh_cheaptrick(ctx, None, b, x_d, tp_d, f0_d, vuv_d, fs, fft, -0.15, spec_d, None))
ok(lib.wh_stream_sync(None))
spec = np.empty((nf, k)); f0_after = np.empty(nf)
ok(lib.wh_memcpy_d2h(spec.ctypes.data_as(vp), spec_d, spec.nbytes, None))
ok(lib.wh_memcpy_d2h(f0_after.ctypes.data_as(vp), f0_d, f0_after.nbytes, None))
flags = (ctypes.c_int32 * 16)()
ok(lib.wh_take_flags(ctx, None, flags))
for p in (x_d, tp_d, f0_d, vuv_d, spec_d):
    ok(lib.wh_free(p))
ok(lib.wh_batch_destroy(b)); ok(lib.wh_ctx_destroy(ctx))
assert "torch" not in sys.modules
np.savez(out_path, spec=spec, f0_after=f0_after, flags=np.array(list(flags)))
'''


@pytest.mark.gpu
def test_cheaptrick_through_plain_c_abi(tmp_path, golden):
    lib = os.path.join(ROOT, "python-world_amd", "lib", "libworld_hip.so")
    out = str(tmp_path / "out.npz")
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", DRIVER, lib, os.path.join(GOLDEN, "golden_syn16k.npz"), out],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = np.load(out)
    g = golden("syn16k")
    assert not res["flags"].any()
    assert rel_rms(res["spec"].T, g["ct_spectrogram"]) < 1e-8
    assert np.array_equal(res["f0_after"], g["ct_f0_after"])
