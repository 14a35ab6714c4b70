"""Pulse-by-pulse overlap-add synthesis — drop-in for world/synthesis.py:21 of the reference, executed
by the HIP kernels behind wh_synthesis (include/world_hip.h)."""
import ctypes
import functools

import numpy as np

from . import _hip


@functools.lru_cache(maxsize=4096)
def _arange_len(start, stop, step):
    return len(np.arange(start, stop, step))


def time_axis_params(temporal_positions, fs):
    """(ny, t0, dt) of np.arange(tp[0], tp[-1] + 1/fs, 1/fs) — evaluated with NumPy itself so that the
    float-arange length quirk is reproduced (memoised per distinct end points: a batch of equal-length
    utterances asks once); dt is the step NumPy actually uses."""
    tp0 = float(temporal_positions[0])
    step = 1 / fs
    ny = _arange_len(tp0, float(temporal_positions[-1]) + step, step)
    dt = (tp0 + step) - tp0
    return ny, tp0, dt


def default_pulse_cap(ny_list):
    """Pulse slots per utterance the batched decode allocates by default: enough for a mean f0 below fs/8."""
    return int(max(ny_list)) // 8 + 64


def safe_pulse_cap(ny_list):
    """Upper bound for any f0 below fs/2 (a pulse needs a phase wrap, i.e. at least two samples)."""
    return int(max(ny_list)) // 2 + 16


def synthesis_timebase_device(rt_tb, batch, tp_d, f0_d, vuv_d, fs, ny_list, t0_list, dt_list, pulse_cap, f0_low_limit=0.0):
    """The half of synthesis() that depends on the time base alone (wh_synthesis_timebase), on ``rt_tb``'s context and
    stream; its results stay in that context's workspace for ``synthesis_device(..., timebase_rt=rt_tb)``.
    ``f0_low_limit`` > 0: ``f0_d`` is the F0 stage's output, read as encode() leaves it after CheapTrick and D4C."""
    y_off = np.concatenate([[0], np.cumsum(ny_list)]).astype(np.int64)
    t0 = np.ascontiguousarray(t0_list, dtype=np.float64)
    dt = np.ascontiguousarray(dt_list, dtype=np.float64)
    vp = ctypes.c_void_p
    _hip.check(rt_tb.lib.wh_synthesis_timebase(rt_tb.ctx, rt_tb.stream(), batch.handle, rt_tb.ptr(tp_d), rt_tb.ptr(f0_d),
                                               rt_tb.ptr(vuv_d), float(fs), y_off.ctypes.data_as(vp),
                                               t0.ctypes.data_as(vp), dt.ctypes.data_as(vp), int(pulse_cap),
                                               float(f0_low_limit)))


def synthesis_device(rt, batch, tp_d, f0_d, vuv_d, spec_d, ap_d, fs, fft_size, ny_list, t0_list, dt_list,
                     noise_d=None, noise_off=None, seed=0, pulse_cap=None, timebase_rt=None):
    """Device-resident core.  spec_d/ap_d are frame-major [F][K].  Returns the concatenated waveform tensor.
    ``timebase_rt``: a runtime whose context already holds this batch's time base (``synthesis_timebase_device`` with
    the same lengths and pulse_cap); only the spectral half (wh_synthesis_render) runs then."""
    y_off = np.concatenate([[0], np.cumsum(ny_list)]).astype(np.int64)
    t0 = np.ascontiguousarray(t0_list, dtype=np.float64)
    dt = np.ascontiguousarray(dt_list, dtype=np.float64)
    if pulse_cap is None:
        pulse_cap = default_pulse_cap(ny_list)  # callers read WH_FLAG_PULSE_OVERFLOW and retry with safe_pulse_cap
    y = rt.empty((int(y_off[-1]),))
    vp = ctypes.c_void_p
    noff = None
    if noise_d is not None:
        noff = np.ascontiguousarray(noise_off, dtype=np.int64)
    noff_p = noff.ctypes.data_as(vp) if noff is not None else vp(None)
    if timebase_rt is not None:
        _hip.check(rt.lib.wh_synthesis_render(rt.ctx, rt.stream(), batch.handle, timebase_rt.ctx, rt.ptr(tp_d),
                                              rt.ptr(spec_d), rt.ptr(ap_d), float(fs), int(fft_size),
                                              y_off.ctypes.data_as(vp), t0.ctypes.data_as(vp), dt.ctypes.data_as(vp),
                                              int(pulse_cap), rt.ptr(noise_d), noff_p, int(seed), rt.ptr(y), vp(None)))
        return y, y_off
    _hip.check(rt.lib.wh_synthesis(rt.ctx, rt.stream(), batch.handle, rt.ptr(tp_d), rt.ptr(f0_d), rt.ptr(vuv_d),
                                   rt.ptr(spec_d), rt.ptr(ap_d), float(fs), int(fft_size), y_off.ctypes.data_as(vp),
                                   t0.ctypes.data_as(vp), dt.ctypes.data_as(vp), int(pulse_cap), rt.ptr(noise_d),
                                   noff_p, int(seed), rt.ptr(y), vp(None)))
    return y, y_off


_PHILOX_SEED_MUL, _PHILOX_UTT_MUL = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03  # philox_key (csrc/wh_synthesis.hip)


def philox_seed_for_offset(seed, first_utt):
    """The seed under which utterance u of a batch draws the noise that utterance ``first_utt + u`` draws under ``seed``
    (the stream key of an utterance is seed * A + u * B + 1 mod 2**64 with A odd, so seed' = seed + first_utt * B / A):
    what lets a batch be decoded in consecutive parts with the noise of the whole."""
    m = 1 << 64
    return (int(seed) + int(first_utt) * _PHILOX_UTT_MUL * pow(_PHILOX_SEED_MUL, -1, m)) % m


def philox_normals(rt, seed, utt, n, q0=0):
    """Samples [q0, q0 + n) of the standard-normal stream that the device-noise decode (``noise=None``) reads for
    utterance ``utt`` of a batch under ``seed`` (wh_philox_normals): a device tensor.  Feeding it back as that
    utterance's ``noise`` reproduces the seeded decode — the hook that makes the Philox path checkable sample by sample
    against the oracle (the stand-in for the reference's np.random.randn draws, world/synthesis.py:93)."""
    out = rt.empty((int(n),))
    _hip.check(rt.lib.wh_philox_normals(rt.ctx, rt.stream(), int(seed), int(utt), int(q0), int(n), rt.ptr(out)))
    return out


def synthesis_plan(rt, batch, tp_d, f0_d, vuv_d, fs, ny_list, t0_list, dt_list, pulse_cap):
    """(pulse counts, exact reference randn draw counts) per utterance."""
    y_off = np.concatenate([[0], np.cumsum(ny_list)]).astype(np.int64)
    t0 = np.ascontiguousarray(t0_list, dtype=np.float64)
    dt = np.ascontiguousarray(dt_list, dtype=np.float64)
    counts = np.zeros(batch.n_utt, dtype=np.int32)
    draws = np.zeros(batch.n_utt, dtype=np.int64)
    vp = ctypes.c_void_p
    _hip.check(rt.lib.wh_synthesis_plan(rt.ctx, rt.stream(), batch.handle, rt.ptr(tp_d), rt.ptr(f0_d), rt.ptr(vuv_d),
                                        float(fs), y_off.ctypes.data_as(vp), t0.ctypes.data_as(vp),
                                        dt.ctypes.data_as(vp), int(pulse_cap), counts.ctypes.data_as(vp),
                                        draws.ctypes.data_as(vp)))
    return counts, draws


@_hip.serialised
def synthesis(source_object, filter_object):
    """Same contract as the reference.  Randomness: exactly as many np.random.randn samples are drawn
    from NumPy's global stream as the reference draws (one randn(max(3, noise_size)) per pulse), so a
    call that starts from the same generator state reproduces the reference's noise and leaves the generator in
    the same state.  Note that the reference's cheaptrick() also consumes the global stream (one rand(K) per frame,
    world/cheaptrick.py:117) and this build's cheaptrick() does not unless ``world.cheaptrick.CONSUME_REFERENCE_RNG``
    is set: to compare seeded encode→decode runs sample for sample, reseed right before decode (or set that flag)."""
    rt = _hip.Runtime.get()
    vuv = np.asarray(source_object['vuv'], dtype=np.float64)
    f0 = np.asarray(source_object['f0'], dtype=np.float64)
    fs = filter_object['fs']
    spectrogram = np.asarray(filter_object['spectrogram'], dtype=np.float64)
    aperiodicity = np.asarray(source_object['aperiodicity'], dtype=np.float64)
    tp = np.asarray(source_object['temporal_positions'], dtype=np.float64)
    nf = _hip.same_frames("synthesis", dense=(("spectrogram", spectrogram), ("aperiodicity", aperiodicity)),
                          temporal_positions=tp, f0=f0, vuv=vuv)
    fft_size = (spectrogram.shape[0] - 1) * 2
    ny, t0, dt = time_axis_params(tp, fs)
    batch = rt.make_batch([0, 0], [0, nf])
    tp_d, f0_d, vuv_d = rt.to_device(tp), rt.to_device(f0), rt.to_device(vuv)
    # (bins, frames) as the reference holds them -> frame-major on the device: uploaded as they lie, transposed there (a
    # strided host copy of the two 3.8 MB arrays was ~2 ms of this call's 3.4)
    spec_d = rt.to_device(spectrogram).transpose(0, 1).contiguous()
    ap_d = rt.to_device(aperiodicity).transpose(0, 1).contiguous()
    cap = safe_pulse_cap([ny])
    counts, draws = synthesis_plan(rt, batch, tp_d, f0_d, vuv_d, fs, [ny], [t0], [dt], cap)
    assert counts[0] > 0  # world/synthesis.py:131
    noise = np.random.randn(int(draws[0]))
    y, _ = synthesis_device(rt, batch, tp_d, f0_d, vuv_d, spec_d, ap_d, fs, fft_size, [ny], [t0], [dt],
                            noise_d=rt.to_device(noise), noise_off=[0, len(noise)], pulse_cap=cap)
    rt.check_flags("synthesis")
    return rt.to_host(y)
