"""Requiem synthesis — drop-in for world/synthesisRequiem.py:12 of the reference, executed by the HIP
kernels behind wh_synthesis_requiem (include/world_hip.h)."""
import ctypes
import weakref

import numpy as np

from . import _hip
from .synthesis import default_pulse_cap, safe_pulse_cap, time_axis_params

_seed_cache = {}  # (id(pulse), id(noise), device) -> (weakref(noise), pulse_d, noise_d)


def seeds_on_device(rt, seeds):
    """Device copies of the seed tables, uploaded once per (tables, device): a decode loop that passes the same
    ``seeds`` dict does not re-upload 0.5-2 MB per call.  The cache entry dies with the host arrays."""
    if 'pulse_d' in seeds and 'noise_d' in seeds:  # already resident (world.get_seeds_signals.get_seeds_signals_device)
        return seeds['pulse_d'], seeds['noise_d'], seeds['pulse_d'].shape, seeds['noise_d'].shape
    pulse, noise = seeds['pulse'], seeds['noise']
    key = (id(pulse), id(noise), rt.index)
    hit = _seed_cache.get(key)
    if hit is not None and hit[0]() is noise:
        return hit[1], hit[2], pulse.shape, noise.shape
    pulse_d = rt.to_device(np.ascontiguousarray(pulse, dtype=np.float64))
    noise_d = rt.to_device(np.ascontiguousarray(noise, dtype=np.float64))
    try:
        _seed_cache[key] = (weakref.ref(noise, lambda _r, k=key: _seed_cache.pop(k, None)), pulse_d, noise_d)
    except TypeError:
        pass
    return pulse_d, noise_d, pulse.shape, noise.shape


def generate_noise(N, noise_seed, frequency_band):
    """Host mirror of the reference helper (world/synthesisRequiem.py:131-141).  The device path does not
    call it; it exists because the reference keeps the circular-read cursor in this function's
    ``current_index`` attribute, and callers reset it there (`generate_noise.current_index = None`)."""
    if np.all(generate_noise.current_index == None):  # noqa: E711
        generate_noise.current_index = np.zeros(noise_seed.shape[1])
    n_len = noise_seed.shape[0]
    start = generate_noise.current_index[frequency_band]
    index = np.remainder(np.arange(start, start + N), n_len).astype(int)
    generate_noise.current_index[frequency_band] = index[-1]
    return noise_seed[index, frequency_band]


generate_noise.current_index = None


def _advance(cursor, ny, noise_len):
    """Cursor after reading ny samples: the reference stores index[-1], not index[-1]+1."""
    return np.remainder(cursor + ny - 1, noise_len)


def seed_table_shape(fs, seeds=None):
    """(noise_length, nb) of the seed tables a decode at ``fs`` reads: those of ``seeds`` or the default ones."""
    if seeds is not None:
        shape = seeds['noise_d'].shape if 'noise_d' in seeds else seeds['noise'].shape
        return int(shape[0]), int(shape[1])
    from .get_seeds_signals import _BAND_STEP, _UPPER
    return int(2 ** np.ceil(np.log2(fs / 2))), int(2 + np.floor(min(_UPPER, fs / 2 - _BAND_STEP) / _BAND_STEP))


def cursor_after(frame_times, fs, cursor, noise_len):
    """The noise-seed read position (nb,) after utterances with the given per-utterance ``frame_times`` arrays have been
    rendered from ``cursor`` one after the other — computed on the host from the output lengths alone: where a decode
    that follows (the next range of a batch cut over several devices, world.pool) has to start to draw what the single
    batch draws."""
    cur = np.array(cursor, dtype=np.float64)
    for tp in frame_times:
        cur = _advance(cur, time_axis_params(tp, fs)[0], noise_len)
    return cur


def synthesis_requiem_core(rt, batch, tp_d, f0_d, vuv_d, spec_d, band_d, fs, fft_size, geo, hops, seeds, cursors,
                           pulse_cap=None):
    """Device-resident core.  geo: [(ny, t0, dt)] per utterance; cursors: (n_utt, nb) start positions."""
    ny = [g[0] for g in geo]
    y_off = np.concatenate([[0], np.cumsum(ny)]).astype(np.int64)
    t0 = np.ascontiguousarray([g[1] for g in geo], dtype=np.float64)
    dt = np.ascontiguousarray([g[2] for g in geo], dtype=np.float64)
    hop = np.ascontiguousarray(hops, dtype=np.int64)
    cur = np.ascontiguousarray(cursors, dtype=np.int64)
    pulse_d, noise_d, pshape, nshape = seeds_on_device(rt, seeds)
    nb = int(pshape[1])
    if pulse_cap is None:
        pulse_cap = default_pulse_cap(ny)  # callers read WH_FLAG_PULSE_OVERFLOW and retry with safe_pulse_cap
    y = rt.empty((int(y_off[-1]),))
    vp = ctypes.c_void_p
    _hip.check(rt.lib.wh_synthesis_requiem(
        rt.ctx, rt.stream(), batch.handle, rt.ptr(tp_d), rt.ptr(f0_d), rt.ptr(vuv_d), rt.ptr(spec_d), rt.ptr(band_d),
        float(fs), int(fft_size), y_off.ctypes.data_as(vp), t0.ctypes.data_as(vp), dt.ctypes.data_as(vp),
        hop.ctypes.data_as(vp), int(pulse_cap), rt.ptr(pulse_d), int(pshape[0]), rt.ptr(noise_d),
        int(nshape[0]), int(nb), cur.ctypes.data_as(vp), rt.ptr(y)))
    return y, y_off


_default_seeds = {}  # (fs, device, lane) -> device-resident seed tables (the batched path's default when the caller passes none)


def synthesis_requiem_device(rt, enc, ny, geo, seeds=None, cursor=None, pulse_cap=None):
    """Batch decode of a BatchEncoding (is_requiem=True).  Utterances consume the noise seed one after the
    other exactly like consecutive reference calls sharing the persistent cursor (world/synthesisRequiem.py:131-141,
    world/main.py:205-206); ``cursor`` (nb,) is the position the first utterance starts at (default zeros)."""
    from .get_seeds_signals import get_seeds_signals_device

    if seeds is None:  # default of the batched path: tables generated on the device, once per (fs, device, lane) — a
        # lane is a context and a stream of its own, possibly driven by a host thread of its own (world.pool): tables
        # made on another lane's stream would be read here without any ordering
        key = (enc.fs, rt.index, rt.lane)
        seeds = _default_seeds.get(key)
        if seeds is None:
            seeds = _default_seeds[key] = get_seeds_signals_device(enc.fs, seed=0, rt=rt)
    _, _, pshape, nshape = seeds_on_device(rt, seeds)
    nb = int(pshape[1])
    nlen = int(nshape[0])
    cur = np.zeros(nb) if cursor is None else np.array(cursor, dtype=np.float64)
    fo = enc.batch.frame_off
    tp_h = enc.host_times()
    hops, cursors = [], []
    for u in range(enc.n_utt):
        t = tp_h[int(fo[u]):int(fo[u + 1])]
        hops.append(int((t[1] - t[0]) * enc.fs))
        cursors.append(cur.copy())
        cur = _advance(cur, ny[u], nlen)
    y, y_off = synthesis_requiem_core(rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, enc.spectrogram,
                                      enc.aperiodicity, enc.fs, enc.fft_size, geo, hops, seeds, np.array(cursors),
                                      pulse_cap=pulse_cap)
    enc.requiem_cursor = cur  # where the next batch would continue (the reference's generate_noise.current_index)
    return y, y_off


@_hip.serialised
def synthesisRequiem(source_object, filter_object, seeds_signals):
    """Same contract as the reference, including the cursor that persists across calls in
    ``generate_noise.current_index``."""
    rt = _hip.Runtime.get()
    fs = filter_object['fs']
    tp = np.asarray(source_object['temporal_positions'], dtype=np.float64)
    f0 = np.asarray(source_object['f0'], dtype=np.float64)
    vuv = np.asarray(source_object['vuv'], dtype=np.float64)
    spectrogram = np.asarray(filter_object['spectrogram'], dtype=np.float64)
    band = np.asarray(source_object['aperiodicity'], dtype=np.float64)
    _hip.same_frames("synthesisRequiem", dense=(("spectrogram", spectrogram), ("aperiodicity", band)),
                     temporal_positions=tp, f0=f0, vuv=vuv)
    fft_size = (spectrogram.shape[0] - 1) * 2
    nb = seeds_signals['pulse'].shape[1]
    nlen = seeds_signals['noise'].shape[0]
    if np.all(generate_noise.current_index == None):  # noqa: E711
        generate_noise.current_index = np.zeros(nb)
    geo = [time_axis_params(tp, fs)]
    batch = rt.make_batch([0, 0], [0, len(tp)])
    y, _ = synthesis_requiem_core(rt, batch, rt.to_device(tp), rt.to_device(f0), rt.to_device(vuv),
                                  rt.to_device(spectrogram).transpose(0, 1).contiguous(),
                                  rt.to_device(band).transpose(0, 1).contiguous(), fs, fft_size, geo,
                                  [int((tp[1] - tp[0]) * fs)], seeds_signals,
                                  np.array([generate_noise.current_index]), pulse_cap=safe_pulse_cap([geo[0][0]]))
    rt.check_flags("synthesisRequiem")
    generate_noise.current_index = _advance(np.asarray(generate_noise.current_index, dtype=np.float64), geo[0][0], nlen)
    return rt.to_host(y)
