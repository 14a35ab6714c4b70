"""DIO F0 estimator — drop-in for world/dio.py:10 of the reference, executed by the HIP kernels
behind wh_dio (include/world_hip.h)."""
import ctypes

import numpy as np

from . import _hip, _tables


def dio_device(rt, batch, x_d, tp_d, fs, f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000,
               frame_period=5, allowed_range=0.1, want_candidates=False, index_bias=None):
    """Device-resident core.  Returns (f0, vuv, f0_candidates or None, raw_f0_candidates or None) as
    flat device tensors laid out per the ABI."""
    tb = _tables.dio_tables(f0_floor, f0_ceil, channels_in_octave, target_fs)
    if index_bias is not None:  # test hook: pin the Nuttall-argmax tie to a recorded fixture
        tb["band_bias"] = np.asarray(index_bias, dtype=np.int32)
    nb = len(tb["band_f0"])
    nf = batch.total_frames
    f0 = rt.empty((nf,))
    vuv = rt.empty((nf,))
    cand = rt.empty((nf * nb,)) if want_candidates else None
    raw = rt.empty((nf * nb,)) if want_candidates else None
    vp = ctypes.c_void_p
    _hip.check(rt.lib.wh_dio(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), float(fs), float(f0_floor),
                             float(f0_ceil), float(target_fs), float(frame_period), float(allowed_range), nb,
                             tb["band_f0"].ctypes.data_as(vp), tb["band_bias"].ctypes.data_as(vp),
                             tb["band_len"].ctypes.data_as(vp), tb["band_taps"].ctypes.data_as(vp),
                             tb["lowcut"].ctypes.data_as(vp), int(tb["lowcut_half"]), rt.ptr(f0), rt.ptr(vuv),
                             rt.ptr(cand), rt.ptr(raw)))
    return f0, vuv, cand, raw


@_hip.serialised
def dio(x, fs, f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000, frame_period=5, allowed_range=0.1,
        _index_bias=None):
    """Same contract as the reference: dict with 'f0', 'f0_candidates' (nb,F), 'raw_f0_candidates' (nb,F),
    'temporal_positions', 'vuv'."""
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    nf = _tables.frame_count(len(x), fs, frame_period)
    tp = _tables.frame_times(nf, frame_period)
    batch = rt.make_batch([0, len(x)], [0, nf])
    f0, vuv, cand, raw = dio_device(rt, batch, rt.to_device(x), rt.to_device(tp), fs, f0_floor, f0_ceil,
                                    channels_in_octave, target_fs, frame_period, allowed_range, want_candidates=True,
                                    index_bias=_index_bias)
    rt.check_flags("dio")
    nb = cand.numel() // nf
    return {'f0': f0.cpu().numpy(),
            'f0_candidates': cand.cpu().numpy().reshape(nb, nf),
            'raw_f0_candidates': raw.cpu().numpy().reshape(nb, nf),
            'temporal_positions': tp,
            'vuv': vuv.cpu().numpy()}
