"""Harvest F0 estimator — drop-in for world/harvest.py:17 of the reference, executed by the HIP kernels
behind wh_harvest (include/world_hip.h)."""
import ctypes

import numpy as np

from . import _hip, _tables


def counted_event_caps(rt):
    """Capacities for a repeat of the last ``harvest_device`` call of ``rt``: what that call counted per (utterance,
    channel) — exact also when it overflowed its lists (WH_FLAG_EVENT_OVERFLOW).  Waits for the stream."""
    caps = np.empty(rt.harvest_lists, dtype=np.int64)
    _hip.check(rt.lib.wh_harvest_event_counts(rt.ctx, rt.stream(), caps.ctypes.data_as(ctypes.c_void_p), len(caps)))
    return caps


def flat_samples(x):
    """Samples of a host waveform that repeat their predecessor exactly (digital silence, clipping plateaus) — what the
    capacity estimate of Harvest's zero-crossing lists cannot see from the length alone (``hinted_event_caps``)."""
    x = np.asarray(x)
    return int(np.count_nonzero(x[1:] == x[:-1]))


def hinted_event_caps(batch, fs, tb):
    """The library's estimate (three times a channel's centre frequency per second + 64) plus what the flat stretches
    of the waveforms (``batch.flat_samples``, set by the callers that see host arrays) can add: where the decimated,
    mean-removed signal is a DC level, the first difference of its filtered image is rounding noise and changes sign up
    to every other sample — one more entry per two flat samples in a train.  None when there is nothing to add (the
    library then sizes the lists itself).  An estimate still: a checked caller repeats a call that exceeds it."""
    flat = getattr(batch, "flat_samples", None)
    r = int(tb["r"])
    if flat is None or max(flat) // (2 * r) <= 32:
        return None
    fs_d = fs / r
    lens = np.diff(batch.x_off)
    caps = np.empty((batch.n_utt, len(tb["band_f0"])), dtype=np.int64)
    for u in range(batch.n_utt):
        ylen = int(lens[u]) // r + 2
        caps[u] = np.ceil(ylen / fs_d * tb["band_f0"] * 3.0).astype(np.int64) + 64 + int(flat[u]) // (2 * r) + 16
    return caps.ravel()


def harvest_device(rt, batch, x_d, tp_d, fs, f0_floor=71, f0_ceil=800, frame_period=5, debug=False, event_caps=None):
    """Device-resident core: returns (f0, vuv) device tensors on the output frame grid
    (plus a dict of debug tensors when ``debug``).

    ``event_caps``: capacities of the zero-crossing lists (the reference's ragged arrays, world/harvest.py:283-297).
    None — the library's estimate (three times a channel's centre frequency per second; with ``batch.flat_samples``
    known, ``hinted_event_caps``); a stretch that is constant up to rounding (digital silence next to signal) can
    exceed it: the call then raises WH_FLAG_EVENT_OVERFLOW, and a checked caller repeats it with ``counted_event_caps``.  ``'safe'`` — the bound no signal exceeds (4.3 x the memory):
    for calls kept asynchronous on material known to hold such stretches.  An int64 array [n_utt * channels] — as given."""
    tb = _tables.harvest_tables(fs, f0_floor, f0_ceil)
    if event_caps is None:
        event_caps = hinted_event_caps(batch, fs, tb)
    if isinstance(event_caps, str):
        if event_caps != 'safe':
            raise ValueError("event_caps: None, 'safe' or an array")
        _hip.check(rt.lib.wh_harvest_set_event_caps(rt.ctx, None, -1))
    elif event_caps is not None:
        caps = np.ascontiguousarray(event_caps, dtype=np.int64)
        _hip.check(rt.lib.wh_harvest_set_event_caps(rt.ctx, caps.ctypes.data_as(ctypes.c_void_p), len(caps)))
    try:
        return _harvest_launch(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period, debug, tb)
    finally:
        if isinstance(event_caps, str):
            _hip.check(rt.lib.wh_harvest_set_event_caps(rt.ctx, None, 0))


def _harvest_launch(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period, debug, tb):
    nf = batch.total_frames
    rt.harvest_lists = batch.n_utt * len(tb["band_f0"])  # (counted_event_caps)
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


@_hip.serialised
def harvest(x, fs, f0_floor=71, f0_ceil=800, frame_period=5):
    """Same contract as the reference: {'temporal_positions', 'f0', 'vuv'} on the frame_period grid."""
    rt = _hip.Runtime.get()
    x = np.asarray(x, dtype=np.float64)
    nf = _tables.frame_count(len(x), fs, frame_period)
    tp = _tables.frame_times(nf, frame_period)
    batch = rt.make_batch([0, len(x)], [0, nf])
    batch.flat_samples = [flat_samples(x)]
    x_d, tp_d = rt.to_device(x), rt.to_device(tp)
    f0, vuv = harvest_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period)
    if rt.check_flags("harvest", allow=(_hip.FLAG_EVENT_OVERFLOW,))[_hip.FLAG_EVENT_OVERFLOW]:
        # stretches constant up to rounding: more crossings than estimated — once more with the counted capacities
        f0, vuv = harvest_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period,
                                 event_caps=counted_event_caps(rt))
        rt.check_flags("harvest")
    return {'temporal_positions': tp, 'f0': f0.cpu().numpy(), 'vuv': vuv.cpu().numpy()}
