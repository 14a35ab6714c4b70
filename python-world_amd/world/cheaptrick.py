"""CheapTrick spectral envelope — drop-in for world/cheaptrick.py:9 of the reference, executed by
the HIP kernel behind wh_cheaptrick (include/world_hip.h)."""
import numpy as np

from . import _hip

# The reference dithers every smoothed spectrum with np.random.rand(K)*eps from NumPy's GLOBAL stream
# (world/cheaptrick.py:117); this build adds the dither's mean eps/2 instead and leaves the generator alone, so a
# seeded reference encode()+decode() and a seeded encode()+decode() here enter synthesis with different generator
# states.  Set this flag to draw (and discard) the same F*K uniforms per call and keep the two generators in step.
CONSUME_REFERENCE_RNG = False


def default_fft_size(fs, f0_low_limit=71):
    return int(2 ** np.ceil(np.log2(3 * fs / f0_low_limit + 1)))


def cheaptrick_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, fft_size, q1=-0.15, want_ps=False):
    """Device-resident core: returns (spectrogram [F][K], ps [F][fft] complex or None); f0_d is updated in place."""
    nf = batch.total_frames
    k = fft_size // 2 + 1
    spec = rt.empty((nf, k))
    ps = rt.empty((nf, fft_size), dtype=rt.torch.complex128) if want_ps else None
    _hip.check(rt.lib.wh_cheaptrick(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d),
                                    rt.ptr(vuv_d), float(fs), int(fft_size), float(q1), rt.ptr(spec), rt.ptr(ps)))
    return spec, ps


@_hip.serialised
def cheaptrick(x, fs, source_object, q1=-0.15, fft_size=None):
    """Same contract as the reference: returns {'temporal_positions','spectrogram' (K,F),'fs',
    'ps spectrogram' (fft,F) complex} and overwrites source_object['f0'] in place with the 500 Hz
    substitutions (world/cheaptrick.py:26-27,32-33)."""
    if fft_size is None:
        fft_size = default_fft_size(fs)
    fft_size = int(fft_size)
    x = np.asarray(x, dtype=np.float64)
    tp = source_object['temporal_positions']
    f0 = source_object['f0']
    nf = _hip.same_frames("cheaptrick", temporal_positions=tp, f0=f0, vuv=source_object['vuv'])
    rt = _hip.Runtime.get()
    batch = rt.make_batch([0, len(x)], [0, nf])
    x_d = rt.to_device(x)
    tp_d = rt.to_device(tp)
    f0_d = rt.to_device(f0)
    vuv_d = rt.to_device(source_object['vuv'])
    spec, ps = cheaptrick_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, fft_size, q1, want_ps=True)
    f0[...] = f0_d.cpu().numpy()  # the reference mutates the caller's array
    if CONSUME_REFERENCE_RNG:
        np.random.rand(nf * (fft_size // 2 + 1))  # one rand(K) per frame in the reference: same stream position
    return {'temporal_positions': tp,
            'spectrogram': rt.to_host(spec, transpose=True),
            'fs': fs,
            'ps spectrogram': rt.to_host(ps, transpose=True)}
