"""D4C aperiodicity ("Love Train") — drop-in for world/d4c.py:10 of the reference, executed by the
HIP kernels behind wh_d4c (include/world_hip.h)."""
import numpy as np

from . import _hip


def d4c_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, threshold, fft_size_for_spectrum, want_coarse=False):
    """Device-resident core: returns (aperiodicity [F][K], coarse_ap [F][nap] or None); f0_d zeroed where vuv==0."""
    nf = batch.total_frames
    interval = 2000 if fs < 16000 else 3000
    nap = int(np.floor(np.min([15000, fs / 2 - interval]) / interval))
    assert nap > 0  # world/d4c.py:35
    k = fft_size_for_spectrum // 2 + 1
    ap = rt.empty((nf, k))
    coarse = rt.empty((nf, nap)) if want_coarse else None
    _hip.check(rt.lib.wh_d4c(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d),
                             rt.ptr(vuv_d), float(fs), float(threshold), int(fft_size_for_spectrum), rt.ptr(ap),
                             rt.ptr(coarse)))
    return ap, coarse


@_hip.serialised
def d4c(x, fs, f0_object, threshold=0.85, fft_size_for_spectrum=None):
    """Same contract as the reference: zeroes f0_object['f0'] where vuv==0, adds 'aperiodicity' (K,F)
    and 'coarse_ap' (nap,F) to the SAME dict and returns it (world/d4c.py:28-32,61-64)."""
    if fft_size_for_spectrum is None:
        fft_size_for_spectrum = int(2 ** np.ceil(np.log2(3 * fs / 71 + 1)))
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    f0 = f0_object['f0']
    _hip.same_frames("d4c", temporal_positions=f0_object['temporal_positions'], f0=f0, vuv=f0_object['vuv'])
    batch = rt.make_batch([0, len(x)], [0, len(f0)])
    f0_d = rt.to_device(f0)
    ap, coarse = d4c_device(rt, batch, rt.to_device(x), rt.to_device(f0_object['temporal_positions']), f0_d,
                            rt.to_device(f0_object['vuv']), fs, threshold, int(fft_size_for_spectrum), want_coarse=True)
    f0[...] = f0_d.cpu().numpy()
    f0_object['aperiodicity'] = rt.to_host(ap, transpose=True)
    f0_object['coarse_ap'] = rt.to_host(coarse, transpose=True)
    return f0_object
