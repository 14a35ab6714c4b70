"""Spectral feature heads of the reference's World class (world/main.py:216-365): mel filterbank, log filterbank
energies, mel-cepstrum, its inverse and context stacking — the step callers run right after encode()
(test/spectralFeatures.py:27-50).  The per-frame products run on the MI355X's FP64 matrix cores behind
wh_feature_matmul (include/world_hip.h); the small tables that define them (filterbank, pre-emphasis response,
warped cosine bases) are built here on the host with the reference's own NumPy / SciPy expressions and handed
through the ABI as data.  No CPU fallback: the *_device functions need the library and a GPU."""
import ctypes
import functools

import numpy as np

from . import _hip


def hz2mel(hz):
    """world/main.py:257-263."""
    return 2595 * np.log10(1 + hz / 700.)


def mel2hz(mel):
    """world/main.py:265-271."""
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """Mel filterbank, one triangular filter per row over nfft/2+1 bins (world/main.py:275-303)."""
    highfreq = highfreq or samplerate / 2
    assert highfreq <= samplerate / 2, "highfreq is greater than samplerate/2"
    edges = np.floor((nfft + 1) * mel2hz(np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        rise = np.arange(int(lo), int(mid))
        fall = np.arange(int(mid), int(hi))
        fbank[j, rise] = (rise - lo) / (mid - lo)
        fbank[j, fall] = (hi - fall) / (hi - mid)
    return fbank


class _Tagged:
    """A cached weight table with its content tag (world._hip.table_tag), computed once when the table is built."""

    def __init__(self, w):
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.w.setflags(write=False)
        self.tag = _hip.table_tag(self.w)


@functools.lru_cache(maxsize=16)
def _lfbank_tables(d, prefac, fs, nfilt, lowfreq, highfreq):
    from scipy.signal import freqz

    nfft = (d - 1) * 2
    _, h = freqz([1, -prefac], [1], d)  # pre-emphasis response on the D bins (main.py:313)
    fb = get_filterbanks(nfilt, nfft, fs, lowfreq, highfreq)
    return np.ascontiguousarray(np.abs(h)), _Tagged(fb.T), 1 / nfft


@functools.lru_cache(maxsize=16)
def _mcep_matrix(d, n0, fs, lowhz, highhz):
    """(D, n0): log-spectrum -> first n0 cepstral coefficients, i.e. the mel-warp gather followed by the first n0
    rows of the inverse real FFT of length 2(D-1) (main.py:330-341).  np.interp at the integer-valued warp positions is
    a gather (clamped at the last bin), so both steps are linear in the log spectrum."""
    n = 2 * (d - 1)
    warp = np.floor((n + 1) * mel2hz(np.linspace(hz2mel(lowhz), hz2mel(highhz), d)) / fs)
    src = np.clip(warp, 0, d - 1).astype(np.int64)
    k = np.arange(d)[:, None]
    m = np.arange(n0)[None, :]
    basis = 2 * np.cos(2 * np.pi * k * m / n) / n       # irfft of a real half spectrum, output sample m
    basis[0, :] = 1.0 / n
    basis[d - 1, :] = np.cos(np.pi * m[0]) / n            # Nyquist bin: (-1)^m
    w = np.zeros((d, n0))
    np.add.at(w, src, basis)
    return _Tagged(w)


@functools.lru_cache(maxsize=16)
def _imcep_matrix(n0, fft_size):
    """(n0, K): cepstrum -> log spectrum on the linear axis: real FFT of the symmetric zero-padded cepstrum, then the
    reference's np.interp from the warped axis (knots floor(fft_size * mel2hz(melpoints) / 16000), main.py:348-357 —
    the 16 kHz and the 0-8000 Hz range are hard-coded there).  np.interp is applied to the identity to obtain the
    operator, so its handling of repeated knots is NumPy's own."""
    k_bins = fft_size // 2 + 1
    k = np.arange(k_bins)[None, :]
    n = np.arange(n0)[:, None]
    cos_rows = 2 * np.cos(2 * np.pi * k * n / fft_size)   # Yc[:, :-n0:-1] mirrors coefficients 1..n0-1
    cos_rows[0, :] = 1.0
    knots = np.floor(fft_size * mel2hz(np.linspace(hz2mel(0), hz2mel(8000), k_bins)) / 16000)
    eye = np.eye(k_bins)
    interp = np.array([np.interp(np.arange(k_bins), knots, row) for row in eye])  # (K source bins, K outputs)
    return _Tagged(cos_rows @ interp)


def feature_matmul_device(rt, a_d, n_rows, ka, lda, w, prologue=0, p=None, pscale=1.0, epilogue=0):
    """out[f][n] = epi(sum_k pro(A[f][k]) * w[k][n]) on the device; a_d is a device tensor, w / p host arrays.
    ``w`` may be a ``_Tagged`` table: the device copy of the padded matrix is then looked up by tag instead of being
    re-padded and compared on every call."""
    tag = 0
    if isinstance(w, _Tagged):
        w, tag = w.w, w.tag
    w = np.ascontiguousarray(w, dtype=np.float64)
    nw = w.shape[1]
    out = rt.empty((int(n_rows), nw))
    vp = ctypes.c_void_p
    pp = np.ascontiguousarray(p, dtype=np.float64).ctypes.data_as(vp) if p is not None else vp(None)
    _hip.check(rt.lib.wh_feature_matmul_tagged(rt.ctx, rt.stream(), rt.ptr(a_d), int(n_rows), int(ka), int(lda),
                                               int(prologue), pp, float(pscale), w.ctypes.data_as(vp), int(nw),
                                               int(epilogue), rt.ptr(out), int(nw), int(tag)))
    return out


def lfbank_device(rt, spec_d, prefac=0.97, fs=16000, nfilt=32, lowfreq=0, highfreq=None):
    """encode_lfbank on a frame-major [F][D] device tensor (BatchEncoding.spectrogram is one)."""
    f, d = spec_d.shape
    absh, fbt, scale = _lfbank_tables(int(d), float(prefac), fs, int(nfilt), lowfreq, highfreq)
    return feature_matmul_device(rt, spec_d, f, d, d, fbt, prologue=1, p=absh, pscale=scale, epilogue=1)


def mcep_device(rt, spec_d, n0=12, fs=16000, lowhz=0, highhz=8000):
    f, d = spec_d.shape
    return feature_matmul_device(rt, spec_d, f, d, d, _mcep_matrix(int(d), int(n0), fs, lowhz, highhz), prologue=2)


def imcep_device(rt, cep_d, fft_size):
    f, n0 = cep_d.shape
    return feature_matmul_device(rt, cep_d, f, n0, n0, _imcep_matrix(int(n0), int(fft_size)), epilogue=2)


def context_device(rt, x_d, w=5):
    n, d = x_d.shape
    out = rt.empty((int(n), (2 * w + 1) * int(d)))
    _hip.check(rt.lib.wh_context_frames(rt.ctx, rt.stream(), rt.ptr(x_d), int(n), int(d), int(w), rt.ptr(out)))
    return out


@_hip.serialised
def _on_device(fn, x, *a, **kw):
    rt = _hip.Runtime.get()
    x_d = rt.to_device(np.ascontiguousarray(x, dtype=np.float64))
    return fn(rt, x_d, *a, **kw).cpu().numpy()


def encode_lfbank(spec, prefac=0.97, fs=16000, nfilt=32, lowfreq=0, highfreq=None):
    """Log mel filterbank energies of an (N frames, D bins) magnitude spectrogram (world/main.py:305-322)."""
    return _on_device(lfbank_device, spec, prefac, fs, nfilt, lowfreq, highfreq)


def encode_mcep(spec, n0=12, fs=16000, lowhz=0, highhz=8000):
    """First n0 mel-cepstral coefficients of an (N, D) magnitude spectrogram (world/main.py:324-341)."""
    return _on_device(mcep_device, spec, n0, fs, lowhz, highhz)


def decode_mcep(cepstrum, fft_size):
    """Magnitude spectrogram (N, fft_size/2+1) from mel-cepstra (world/main.py:343-358)."""
    return _on_device(imcep_device, cepstrum, fft_size)


def get_context(x, w=5):
    """Rows i-w..i+w of X side by side, edges replicated (world/main.py:360-365)."""
    return _on_device(context_device, x, w)
