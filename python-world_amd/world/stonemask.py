"""StoneMask F0 refinement — drop-in for world/stonemask.py:8 of the reference, executed by the HIP
kernel behind wh_stonemask (include/world_hip.h)."""
import ctypes
import math

import numpy as np

from . import _hip, _tables


def stonemask_device(rt, batch, x_d, tp_d, f0_d, fs, min_f0):
    """Device-resident core: returns a NEW refined-f0 device tensor.  ``min_f0`` bounds the longest
    analysis window (lowest non-zero f0 that can occur, e.g. DIO's f0_floor)."""
    kmax = int(math.ceil(3 * fs / min_f0 / 2))
    # the bound is rounded up to a multiple of 32: the library caches a window table per (rate, bound) — up to 2 MB each —
    # and World.encode / stonemask() derive the bound from the contour's own minimum, another one for every utterance
    # (a frame's window depends on its f0 alone: a longer bound changes no result)
    kmax = -(-kmax // 32) * 32
    qt = _tables.quantised_times(fs, kmax)
    out = rt.empty((batch.total_frames,))
    _hip.check(rt.lib.wh_stonemask(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d),
                                   float(fs), qt.ctypes.data_as(ctypes.c_void_p), kmax, rt.ptr(out)))
    return out


@_hip.serialised
def stonemask(x, fs, temporal_positions, f0):
    """Same contract as the reference: returns a new refined f0 array; the input is not modified."""
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(f0, dtype=np.float64)
    _hip.same_frames("stonemask", temporal_positions=temporal_positions, f0=f0)
    rt = _hip.Runtime.get()
    pos = f0[f0 > 0]
    if len(pos) == 0:
        return np.copy(f0)
    batch = rt.make_batch([0, len(x)], [0, len(f0)])
    out = stonemask_device(rt, batch, rt.to_device(x), rt.to_device(temporal_positions), rt.to_device(f0), fs,
                           float(pos.min()))
    rt.check_flags("stonemask")
    return out.cpu().numpy()
