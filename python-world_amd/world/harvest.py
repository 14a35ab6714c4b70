"""Harvest F0 estimator — drop-in for world/harvest.py:17 of the reference, executed by the HIP kernels
behind wh_harvest (include/world_hip.h)."""
import ctypes

import numpy as np

from . import _hip, _tables


def harvest_device(rt, batch, x_d, tp_d, fs, f0_floor=71, f0_ceil=800, frame_period=5, debug=False):
    """Device-resident core: returns (f0, vuv) device tensors on the output frame grid
    (plus a dict of debug tensors when ``debug``)."""
    tb = _tables.harvest_tables(fs, f0_floor, f0_ceil)
    nf = batch.total_frames
    f0 = rt.empty((nf,))
    vuv = rt.empty((nf,))
    vp = ctypes.c_void_p
    dbg = {}
    dy = draw = d1 = None
    if debug:
        lens = np.diff(batch.x_off)
        nf1 = [int(1000 * n / fs / 1 + 1) for n in lens]
        dy = rt.zeros((int(sum(lens)),))  # upper bound on the decimated length
        draw = rt.zeros((int(sum(nf1)) * len(tb["band_f0"]),))
        d1 = rt.zeros((int(sum(nf1)),))
        dbg = {"y": dy, "raw": draw, "f0_1ms": d1, "nf1": nf1}
    _hip.check(rt.lib.wh_harvest(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), float(fs),
                                 float(f0_floor), float(f0_ceil), float(frame_period), int(tb["r"]),
                                 tb["ba"].ctypes.data_as(vp), tb["zi"].ctypes.data_as(vp), len(tb["band_f0"]),
                                 tb["band_f0"].ctypes.data_as(vp), tb["band_half"].ctypes.data_as(vp),
                                 tb["band_taps"].ctypes.data_as(vp), rt.ptr(f0), rt.ptr(vuv), rt.ptr(dy), rt.ptr(draw),
                                 rt.ptr(d1)))
    if debug:
        return f0, vuv, dbg
    return f0, vuv


def harvest(x, fs, f0_floor=71, f0_ceil=800, frame_period=5):
    """Same contract as the reference: {'temporal_positions', 'f0', 'vuv'} on the frame_period grid."""
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    nf = _tables.frame_count(len(x), fs, frame_period)
    tp = _tables.frame_times(nf, frame_period)
    batch = rt.make_batch([0, len(x)], [0, nf])
    f0, vuv = harvest_device(rt, batch, rt.to_device(x), rt.to_device(tp), fs, f0_floor, f0_ceil, frame_period)
    rt.check_flags("harvest")
    return {'temporal_positions': tp, 'f0': f0.cpu().numpy(), 'vuv': vuv.cpu().numpy()}
