#!/usr/bin/env python3
"""Drives the stage functions of the drop-in API (dio, stonemask, harvest, cheaptrick, d4c, d4cRequiem, synthesis,
synthesisRequiem, through World.encode / decode) WITHOUT PyTorch: device memory through the C-ABI's own wh_malloc /
wh_memcpy_* (include/world_hip.h), a minimal stand-in for world._hip.Runtime.  For the sanitizer run
(tools/asan_probe.sh): with torch out of the process the host ASan runtime can be preloaded (torch's HIP start-up
segfaults under it), so that neither the late-link false positives (new / delete resolved to different runtimes) nor
pytest's fd capture stand between a finding and its report.  WH_LIB selects the library (the ASan build)."""
import contextlib
import ctypes
import os
import sys
import types

import numpy as np

sys.modules["torch"] = None  # `import torch` fails from here on: world._hip.load_library tries it to share torch's HIP runtime
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
from world import _hip  # noqa: E402

vp = ctypes.c_void_p
_DT = {"float64": np.float64, "complex128": np.complex128, "int16": np.int16, "int32": np.int32, "int64": np.int64,
       "uint8": np.uint8}


class DevArr:
    """A device buffer with the handful of tensor methods the stage shims use."""

    def __init__(self, rt, shape, dtype):
        self.rt, self.shape, self.dtype = rt, tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = vp()
        _hip.check(rt.lib.wh_malloc(ctypes.byref(p), max(self.nbytes, 8)))
        self.p = p

    def data_ptr(self):
        return self.p.value

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def cpu(self):
        return self

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _hip.check(self.rt.lib.wh_stream_sync(None))
        if self.nbytes:
            _hip.check(self.rt.lib.wh_memcpy_d2h(out.ctypes.data_as(vp), self.p, self.nbytes, None))
            _hip.check(self.rt.lib.wh_stream_sync(None))
        return out

    def transpose(self, a, b):
        return self.rt.to_device(np.ascontiguousarray(self.numpy().swapaxes(a, b)), dtype=self.dtype)

    def contiguous(self):
        return self

    def clone(self):
        return self.rt.to_device(self.numpy(), dtype=self.dtype)

    def __del__(self):
        try:
            self.rt.lib.wh_free(self.p)
        except Exception:
            pass


class NoTorchRuntime:
    index, lane, own_stream = 0, 0, None
    torch = types.SimpleNamespace(**{k: v for k, v in _DT.items()})

    def __init__(self):
        self.lib = _hip.load_library()
        self.lib.wh_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        self.lib.wh_free.argtypes = [vp]
        self.lib.wh_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t, vp]
        self.lib.wh_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t, vp]
        self.lib.wh_memset.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
        self.lib.wh_stream_sync.argtypes = [vp]
        h = vp()
        _hip.check(self.lib.wh_ctx_create(0, ctypes.byref(h)))
        self.ctx = h
        self.device = "hip:0"

    def on_stream(self):
        return contextlib.nullcontext()

    def stream(self):
        return vp(None)

    def to_device(self, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = DevArr(self, a.shape, a.dtype)
        if a.nbytes:
            _hip.check(self.lib.wh_memcpy_h2d(d.p, a.ctypes.data_as(vp), a.nbytes, None))
            _hip.check(self.lib.wh_stream_sync(None))
        return d

    def to_host(self, t, transpose=False):
        a = t.numpy()
        return np.ascontiguousarray(a.T) if transpose else a

    def empty(self, shape, dtype=None):
        return DevArr(self, shape if isinstance(shape, (tuple, list)) else (shape,), dtype or np.float64)

    def zeros(self, shape, dtype=None):
        d = self.empty(shape, dtype)
        if d.nbytes:
            _hip.check(self.lib.wh_memset(d.p, 0, d.nbytes, None))
        return d

    @staticmethod
    def ptr(t):
        return vp(t.data_ptr()) if t is not None else vp(None)

    take_flags = _hip.Runtime.take_flags
    check_flags = _hip.Runtime.check_flags
    raise_for_flags = staticmethod(_hip.Runtime.raise_for_flags)

    def make_batch(self, x_off, frame_off):
        return _hip.Batch(self, x_off, frame_off)


def main():
    """usage: notorch_harness.py [fs method requiem]  (default: every combination in one process)"""
    rt = NoTorchRuntime()
    _hip.Runtime.get = classmethod(lambda cls, device_index=None, lane=0: rt)
    from world import main as wmain
    from world._synthetic import synth_utterance

    W = wmain.World()
    if len(sys.argv) > 3:
        combos = [(int(sys.argv[1]), sys.argv[2], sys.argv[3] == "1")]
    else:
        combos = [(fs, m, r) for fs in (16000, 22050, 48000) for m, r in (("dio", False), ("harvest", True), ("harvest", False))]
    done = []
    for fs, method, requiem in combos:
        x = synth_utterance(900 + fs // 1000, fs, {16000: 1.2, 22050: 0.7, 48000: 0.6}.get(fs, 0.8))
        dat = W.encode(fs, x, f0_method=method, is_requiem=requiem)
        print("encoded", (fs, method, requiem), flush=True)
        np.random.seed(1)
        out = W.decode(dict(dat))["out"]
        assert np.all(np.isfinite(out)) and np.all(np.isfinite(dat["spectrogram"])) and len(out) > 0
        done.append((fs, method, requiem, len(dat["f0"]), int(np.sum(dat["vuv"] > 0))))
        print("ok", done[-1], flush=True)
    assert sys.modules.get("torch") is None
    print("HARNESS OK: %d encode+decode passes without torch" % len(done))


if __name__ == "__main__":
    main()
