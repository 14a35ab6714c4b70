"""D4C-Requiem band aperiodicity — drop-in for world/d4cRequiem.py:9 of the reference, executed by
the HIP kernels behind wh_d4c_requiem (include/world_hip.h)."""
import numpy as np

from . import _hip


def d4c_requiem_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, threshold=0.85, fft_size=None):
    """Device-resident core: band aperiodicity in dB, [F][nap+2]; f0_d zeroed where vuv==0."""
    nap = int(np.floor(np.min([15000, fs / 2 - 3000]) / 3000))
    assert nap > 0  # world/d4cRequiem.py:21
    band = rt.empty((batch.total_frames, nap + 2))
    _hip.check(rt.lib.wh_d4c_requiem(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d),
                                     rt.ptr(vuv_d), float(fs), float(threshold), int(fft_size or 0), rt.ptr(band)))
    return band


@_hip.serialised
def d4cRequiem(x, fs, f0_object, threshold=0.85, fft_size=None):
    """Same contract as the reference: zeroes f0 where vuv==0, stores 'aperiodicity' (nap+2,F) in dB in
    the SAME dict and returns it (world/d4cRequiem.py:17-18,42-44)."""
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    f0 = f0_object['f0']
    _hip.same_frames("d4cRequiem", temporal_positions=f0_object['temporal_positions'], f0=f0, vuv=f0_object['vuv'])
    batch = rt.make_batch([0, len(x)], [0, len(f0)])
    f0_d = rt.to_device(f0)
    band = d4c_requiem_device(rt, batch, rt.to_device(x), rt.to_device(f0_object['temporal_positions']), f0_d,
                              rt.to_device(f0_object['vuv']), fs, threshold, fft_size)
    f0[...] = f0_d.cpu().numpy()
    f0_object['aperiodicity'] = rt.to_host(band, transpose=True)
    return f0_object
