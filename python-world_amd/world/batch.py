"""Batched, device-resident WORLD pipeline (not in the reference API: it is the harness the benchmark
configs need — SURVEY.md §8(b) "World.encode_batch/decode_batch").

Utterances are concatenated into one waveform tensor; every stage runs ONE launch per kernel over all
frames x utterances.  Results stay in HBM as torch tensors (frame-major spectrogram / aperiodicity);
`BatchEncoding.to_dicts()` materialises reference-layout NumPy dicts on demand.
"""
import os

import numpy as np

from . import _hip, _tables
from .cheaptrick import cheaptrick_device, default_fft_size
from .d4c import d4c_device
from .d4cRequiem import d4c_requiem_device
from .dio import dio_device
from .stonemask import stonemask_device
from .synthesis import (default_pulse_cap, safe_pulse_cap, synthesis_device, synthesis_timebase_device,
                        time_axis_params)


class _Pending:
    """Placeholder stored under a dense key of an EncodingDict whose value still lives in HBM only."""
    __slots__ = ()

    def __repr__(self):
        return "<resident in HBM: materialised on first access>"


_PENDING = _Pending()


class EncodingDict(dict):
    """One utterance's encode() dict out of a batch (World.encode_batch / BatchEncoding.to_dicts(lazy=True)) — a real
    ``dict`` with the reference's keys and (bins, frames) layouts (world/main.py:144-152) whose DENSE values
    ('spectrogram', 'aperiodicity', 'ps spectrogram') stay on the GPU until somebody reads them: the first
    ``d['spectrogram']`` (or ``get`` / ``items`` / ``values`` / ``dict(d)`` / ``copy`` / pickling) downloads and
    transposes that utterance's slice and stores the NumPy array in the dict, from then on it is an ordinary entry.
    The per-frame scalars (temporal_positions, f0, vuv) are host arrays from the start — they are what scale_pitch /
    scale_duration / modify_duration edit in place.

    World.decode_batch looks at each dict: a dense value that was never handed out cannot have been edited, so its rows
    are taken from the resident encoding (a device slice, no PCIe); one that was read (``np.asarray``, an in-place
    ``dat['spectrogram'][...] = v``, ``warp_spectrum``) or replaced is uploaded from the host like any caller-built
    array.  An ``encode_batch -> scale_pitch -> scale_duration -> decode_batch`` caller therefore moves the waveforms,
    the per-frame scalars and the audio, and nothing else (SURVEY §7.2 "materialise NumPy lazily").

    The dicts of one batch share the resident encoding and keep it alive (8 KB of HBM per frame) until the last of
    them is dropped."""

    DENSE = {'spectrogram': 'spectrogram', 'aperiodicity': 'aperiodicity', 'ps spectrogram': 'ps_spectrogram'}

    def __init__(self, small, enc, utt, dense_keys, order):
        dict.__init__(self)
        self._enc, self._utt = enc, utt
        for k in order:
            dict.__setitem__(self, k, _PENDING if k in dense_keys else small[k])

    # ---- what decode_batch asks --------------------------------------------------------------------------------
    def resident_rows(self, key, rt):
        """Frame-major device rows of a dense value that has not left the GPU (None once it has been read or replaced,
        or when ``rt`` is another device's runtime)."""
        if dict.get(self, key) is not _PENDING or self._enc.rt.index != rt.index:
            return None
        fo = self._enc.batch.frame_off
        return getattr(self._enc, self.DENSE[key])[int(fo[self._utt]):int(fo[self._utt + 1])]

    def _materialise(self, key):
        rows = self.resident_rows(key, self._enc.rt)
        with self._enc.rt.on_stream():
            val = self._enc.rt.to_host(rows, transpose=True)
        dict.__setitem__(self, key, val)
        return val

    # ---- dict protocol: every way of reading a value goes through __getitem__ ---------------------------------------
    def __getitem__(self, key):
        val = dict.__getitem__(self, key)
        return self._materialise(key) if val is _PENDING else val

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):  # (overridden on purpose: dict(d) / {**d} then copy through keys() + __getitem__)
        return dict.__iter__(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def copy(self):
        return dict(self.items())

    def pop(self, key, *default):
        if key in self:
            val = self[key]
            dict.__delitem__(self, key)
            return val
        if default:
            return default[0]
        raise KeyError(key)

    def popitem(self):
        key = next(reversed(self))
        return key, self.pop(key)

    def setdefault(self, key, default=None):
        if key not in self:
            dict.__setitem__(self, key, default)
        return self[key]

    def update(self, *args, **kw):
        for k, v in dict(*args, **kw).items():
            self[k] = v

    def __eq__(self, other):
        return dict(self.items()) == other

    __hash__ = None

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        return "EncodingDict(%s)" % dict.__repr__(self)

    def __reduce__(self):
        return (dict, (self.items(),))


class BatchEncoding:
    """Result of WorldBatch.encode_device: the reference's encode() dict (world/main.py:144-152) for a whole batch,
    resident in HBM.  ``temporal_positions`` / ``f0`` / ``vuv`` are flat per-frame tensors (batch frame layout),
    ``spectrogram`` / ``aperiodicity`` are frame-major [F][K] (aperiodicity [F][nap+2] dB when ``is_requiem``).

    decode_device derives the output geometry (sample counts, time axis, Requiem hop) on the HOST from the frame
    times with NumPy's own float-arange semantics; the host copy those need is cached in
    ``tp_host`` and kept in step by the modifiers below.  Assigning a new tensor to ``temporal_positions`` drops
    the cache (the next decode downloads the frame times); editing the tensor in place behind the object's back
    is not supported — use scale_duration / modify_duration."""

    def __init__(self, rt, batch, fs, tp, f0, vuv, spectrogram, aperiodicity, fft_size, is_requiem, frame_period,
                 tp_host=None, ps_spectrogram=None):
        self.rt, self.batch, self.fs = rt, batch, fs
        # CheapTrick's complex pitch-synchronous spectra, frame-major [F][fft_size] complex128 — encode()'s
        # 'ps spectrogram' (world/main.py:149, world/cheaptrick.py:30,38): kept only when encode_device(want_ps=True)
        self.ps_spectrogram = ps_spectrogram
        self._tp = tp
        self.tp_host = tp_host  # host copy of the frame times: no D2H in decode
        self.f0, self.vuv = f0, vuv
        self.spectrogram, self.aperiodicity = spectrogram, aperiodicity
        self.fft_size, self.is_requiem, self.frame_period = fft_size, is_requiem, frame_period
        self._timebase = None  # synthesis time base computed ahead by WorldBatch.encode_device (see timebase_for)

    @classmethod
    def from_dicts(cls, rt, dats):
        """Upload a list of encode() dicts (reference layout: (bins, frames) arrays) that share fs / is_requiem / FFT
        size as one resident batch.  Dense values of ``EncodingDict``s (World.encode_batch) that never left the GPU
        are taken from their resident encoding instead (device slices; the whole tensor as it is when the list is
        that encoding's utterances in order)."""
        torch = rt.torch
        nfs = [len(d['f0']) for d in dats]
        frame_off = np.concatenate([[0], np.cumsum(nfs)])
        batch = rt.make_batch(np.zeros(len(dats) + 1, dtype=np.int64), frame_off)
        # the three per-frame scalar arrays travel as ONE upload through a pinned staging block (three pageable copies
        # cost three round trips — and a pageable copy of this size waits for whatever another pipeline has in flight on
        # the device: 6.5 ms behind the other half's decode, measured)
        tp_parts = [np.asarray(d['temporal_positions'], dtype=np.float64) for d in dats]
        tp_h = np.concatenate(tp_parts)
        scal_parts = tp_parts + [np.asarray(d[k], dtype=np.float64) for k in ('f0', 'vuv') for d in dats]

        def resident(d, key):
            return d.resident_rows(key, rt) if isinstance(d, EncodingDict) else None

        def rows(key):
            # (bins, frames) per utterance on the host -> one frame-major tensor: uploaded as they lie, transposed on the
            # device (strided host copies of the 64 x 10 s batch took as long as its whole decode)
            parts = [resident(d, key) for d in dats]
            first = dats[0] if isinstance(dats[0], EncodingDict) else None
            if first is not None and all(p is not None for p in parts):
                src = getattr(first._enc, EncodingDict.DENSE[key])
                if (all(isinstance(d, EncodingDict) and d._enc is first._enc and d._utt == u for u, d in enumerate(dats))
                        and src.shape[0] == int(frame_off[-1])):
                    return src  # the encoding's own tensor: nothing is copied
            parts = [p if p is not None else rt.to_device(np.asarray(d[key], dtype=np.float64)).transpose(0, 1)
                     for p, d in zip(parts, dats)]
            return torch.cat(parts, dim=0).contiguous()

        d0 = dats[0]
        if isinstance(d0, EncodingDict) and resident(d0, 'spectrogram') is not None:
            fft_size = d0._enc.fft_size  # (asking the dict for its spectrogram's shape would download it)
        else:
            fft_size = (d0['spectrogram'].shape[0] - 1) * 2
        with rt.on_stream():
            scal = rt.to_device_concat(scal_parts).view(3, -1)
            return cls(rt, batch, d0['fs'], scal[0], scal[1], scal[2], rows('spectrogram'),
                       rows('aperiodicity'), fft_size, bool(d0['is_requiem']), None, tp_host=tp_h)

    @property
    def temporal_positions(self):
        return self._tp

    @temporal_positions.setter
    def temporal_positions(self, value):
        self._tp = value
        self._timebase = None
        self.tp_host = None  # whoever replaces the tensor owns its content: refreshed from the device on demand

    def host_times(self):
        """Frame times on the host (downloaded once if the cache was invalidated)."""
        if self.tp_host is None:
            self.tp_host = self._tp.cpu().numpy()
        return self.tp_host

    @property
    def n_utt(self):
        return self.batch.n_utt

    def _stamp(self):
        """What a prefetched time base was computed from: the f0 / vuv / frame-time tensors and their in-place version
        counters (edits behind the object's back bump them; assigning a new tensor changes the pointer; the modifiers
        below drop the time base themselves).  None where torch keeps no version counter (tensors made under
        torch.inference_mode()): no time base is prefetched or reused then."""
        try:
            return tuple((t.data_ptr(), t._version) for t in (self.f0, self.vuv, self._tp))
        except RuntimeError:
            return None

    def timebase_for(self, owner, pulse_cap):
        """The prefetched time base if it still describes this encoding (same tensors, untouched since encode, made by
        ``owner``'s latest prefetch, same pulse capacity — None: the default one the prefetch used), else None."""
        tb = self._timebase
        # the time-base context is shared by every WorldBatch of a (device, lane): its generation counts the prefetches
        if tb is None or tb["rt"] is not owner._tb_rt or tb["generation"] != tb["rt"].timebase_generation:
            return None
        stamp = self._stamp()
        if stamp is None or tb["stamp"] != stamp or (pulse_cap is not None and tb["pulse_cap"] != pulse_cap):
            return None
        return tb

    def scale_pitch(self, factor):
        """world/main.py:154-162, on the device."""
        self._timebase = None  # (the version counter would say so too; a freed tensor's address can come back)
        self.f0 *= factor
        return self

    def scale_duration(self, factor):
        """world/main.py:170-178, on the device."""
        self._timebase = None
        self._tp *= factor
        if self.tp_host is not None:
            self.tp_host = self.tp_host * factor
        return self

    def warp_spectrum(self, factor):
        """world/main.py:191-196 on the device, in place: every frame becomes np.interp((k/K)**factor, k/K, frame)."""
        import ctypes

        k = self.spectrogram.shape[1]
        src, dx, den = _tables.warp_tables(int(k), float(factor))
        vp = ctypes.c_void_p
        _hip.check(self.rt.lib.wh_warp_spectrum(self.rt.ctx, self.rt.stream(), self.rt.ptr(self.spectrogram),
                                                int(self.spectrogram.shape[0]), int(k), src.ctypes.data_as(vp),
                                                dx.ctypes.data_as(vp), den.ctypes.data_as(vp)))
        return self

    def modify_duration(self, from_time, to_time):
        """world/main.py:180-189 for every utterance of the batch (each with its own last frame time as the final
        anchor): temporal_positions <- np.interp(tp, [0, *from_time, end], to_time), a trailing -1 in to_time meaning
        `end`.  Evaluated by a kernel on the resident frame times; like the reference it installs a NEW array."""
        import ctypes

        from_time = np.asarray(from_time, dtype=np.float64)
        to_time = np.array(to_time, dtype=np.float64)
        assert np.all(np.diff(from_time)) > 0 and np.all(np.diff(to_time)) > 0  # the reference's checks, as written
        assert from_time[0] > 0
        assert len(to_time) == len(from_time) + 2
        tp_h = self.host_times()
        fo = self.batch.frame_off
        xp, fp = [], []
        for u in range(self.n_utt):
            end = float(tp_h[int(fo[u + 1]) - 1])
            assert from_time[-1] < end
            xp.append(np.r_[0, from_time, end])
            t = to_time.copy()
            if t[-1] == -1:
                t[-1] = end
            fp.append(t)
        xp = np.ascontiguousarray(xp, dtype=np.float64)
        fp = np.ascontiguousarray(fp, dtype=np.float64)
        out = self.rt.empty((self.batch.total_frames,))
        vp = ctypes.c_void_p
        _hip.check(self.rt.lib.wh_modify_duration(self.rt.ctx, self.rt.stream(), self.batch.handle, self.rt.ptr(self._tp),
                                                  self.rt.ptr(out), xp.ctypes.data_as(vp), fp.ctypes.data_as(vp),
                                                  int(xp.shape[1])))
        self.temporal_positions = out  # (drops the host cache: decode downloads the new frame times once)
        return self

    # ---- spectral feature heads on the resident spectrogram (world/main.py:305-341; world/features.py) ----------
    def lfbank(self, prefac=0.97, nfilt=32, lowfreq=0, highfreq=None):
        """encode_lfbank of every frame of the batch: device tensor [F][nfilt]."""
        from .features import lfbank_device
        return lfbank_device(self.rt, self.spectrogram, prefac, self.fs, nfilt, lowfreq, highfreq)

    def mcep(self, n0=12, lowhz=0, highhz=8000):
        """encode_mcep of every frame of the batch: device tensor [F][n0]."""
        from .features import mcep_device
        return mcep_device(self.rt, self.spectrogram, n0, self.fs, lowhz, highhz)

    def to_dicts(self, want_ps=False, lazy=False):
        """List of per-utterance dicts with the reference's keys and (bins, frames) layouts.  ``want_ps``: include
        encode()'s 'ps spectrogram' (fft_size, frames) complex128 (world/main.py:149) — the encoding must have been made
        with ``want_ps=True`` (16 B x fft_size per frame: 2 GB for the 64 x 10 s batch, which is why it is opt-in).
        ``lazy``: ``EncodingDict``s — the dense values are downloaded when first read, and World.decode_batch takes the
        ones nobody read straight from this encoding (which the dicts keep alive)."""
        fo = self.batch.frame_off
        with self.rt.on_stream():  # (one download instead of three round trips)
            tp, f0, vuv = self.rt.torch.stack([self.temporal_positions, self.f0, self.vuv]).cpu().numpy()
        if want_ps and self.ps_spectrogram is None:
            raise ValueError("this encoding holds no 'ps spectrogram': encode with want_ps=True")
        # the dense tensors are frame-major on the device; the reference's layout is (bins, frames): every utterance's slice
        # is transposed ON THE DEVICE and lands in pinned host memory (Runtime.to_host) — a pageable download of the whole
        # batch followed by strided host copies took 0.43 s for the 64 x 10 s batch (1.05 GB), this 0.03 - 0.16 s
        to_rows = lambda t, s: self.rt.to_host(t[s], transpose=True)  # noqa: E731
        order = ['temporal_positions', 'vuv', 'fs', 'f0', 'aperiodicity', 'spectrogram', 'is_requiem']
        dense = {'aperiodicity', 'spectrogram'}
        if want_ps:
            order.append('ps spectrogram')
            dense.add('ps spectrogram')
        out = []
        with self.rt.on_stream():
            for u in range(self.n_utt):
                s = slice(int(fo[u]), int(fo[u + 1]))
                small = {'temporal_positions': tp[s].copy(), 'vuv': vuv[s].copy(), 'fs': self.fs, 'f0': f0[s].copy(),
                         'is_requiem': self.is_requiem}
                if lazy:
                    out.append(EncodingDict(small, self, u, dense, order))
                    continue
                out.append({'temporal_positions': small['temporal_positions'], 'vuv': small['vuv'], 'fs': self.fs,
                            'f0': small['f0'], 'aperiodicity': to_rows(self.aperiodicity, s),
                            'spectrogram': to_rows(self.spectrogram, s), 'is_requiem': self.is_requiem})
                if want_ps:
                    out[-1]['ps spectrogram'] = to_rows(self.ps_spectrogram, s)
        return out


SWIPE_DT = 0.005  # swipe()'s default dt, the only one World.encode ever uses (world/main.py:134-135)


def _require_swipe_period(frame_period):
    """A caller-supplied batch grid must be swipe()'s own 5 ms grid (WorldBatch.encode builds it whatever
    ``frame_period`` says, like the reference: world/main.py:134-135)."""
    if frame_period != 5:
        raise ValueError("f0_method='swipe' runs on swipe()'s 5 ms grid (the reference ignores frame_period there); "
                         "got a batch grid with frame_period=%r" % (frame_period,))


def _on_lane_stream(fn):
    """Run a WorldBatch method with its lane's HIP stream as torch's current stream."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with self.rt.lock, self.rt.on_stream():
            return fn(self, *a, **kw)
    return wrapped


class WorldBatch:
    def __init__(self, device_index=None, lane=0, prefetch_timebase=True):
        self.rt = _hip.Runtime.get(device_index, lane)
        # encode_device computes the decode's time base (everything synthesis derives from f0 / vuv / frame times: the
        # exact phase scan, pulse positions, per-pulse frame pairs) on a second stream and context while CheapTrick and
        # D4C run on the first; decode_device then only renders.  Off: decode computes it in line.
        self.prefetch_timebase = prefetch_timebase and os.environ.get("WH_PREFETCH_TIMEBASE", "1") != "0"
        self._tb_rt = None

    def _timebase_runtime(self):
        if self._tb_rt is None:
            self._tb_rt = _hip.Runtime.get(self.rt.index, 1000 + self.rt.lane)  # its own context, workspace and stream
            if not hasattr(self._tb_rt, "timebase_generation"):
                self._tb_rt.timebase_generation = 0
        return self._tb_rt

    def _prefetch_timebase(self, batch, tp_d, tp_host, f0_d, vuv_d, fs, ct_fft):
        """Fork: the time base of the pulse-wise decode from the F0 stage's output, behind everything enqueued so far on
        the lane's stream, on the time-base runtime's stream.  f0 is read through the rule that CheapTrick and D4C will
        apply to it (wh_synthesis_timebase, f0_low_limit), from a private copy (those two kernels rewrite f0 in place)."""
        torch = self.rt.torch
        tb = self._timebase_runtime()
        fo = batch.frame_off
        geo = [time_axis_params(tp_host[int(fo[u]):int(fo[u + 1])], fs) for u in range(batch.n_utt)]
        ny = [g[0] for g in geo]
        cap = default_pulse_cap(ny)
        f0_copy = f0_d.clone()
        main = torch.cuda.current_stream(self.rt.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(tb.own_stream):
            tb.own_stream.wait_event(ready)
            # Conditions still pending in this context (d_flags; everything raised since the last post, in stream order)
            # belong to the previous prefetch.  If a decode rendered from that time base its caller is still owed
            # them (check=False: until WorldBatch.check(); check='deferred': until the next poll): publish them,
            # unread, so that the memset of the coming take/poll cycle cannot lose them — the next
            # _deferred_begin / check() reports them.  If nobody used it (an encoding that was never decoded, or only
            # in line), its conditions are moot and would be blamed on this batch: cleared unpublished.  Never polled
            # here: a poll would drain what an earlier decode_device(check='deferred') has published for its caller.
            tb.post_flags(discard=not getattr(tb, "timebase_consumed", False))
            tb.timebase_consumed = False
            synthesis_timebase_device(tb, batch, tp_d, f0_copy, vuv_d, fs, ny, [g[1] for g in geo], [g[2] for g in geo],
                                      cap, f0_low_limit=fs * 3.0 / (ct_fft - 3.0))
            done = torch.cuda.Event()
            done.record(tb.own_stream)
        tb.timebase_generation += 1
        return {"generation": tb.timebase_generation, "rt": tb, "done": done, "geo": geo, "pulse_cap": cap,
                "keep": (f0_copy,), "stamp": None}

    @_on_lane_stream
    def upload(self, xs, fs, frame_period=5, swipe_grid=False):
        """Concatenate and upload a list of 1-D waveforms (or a 2-D array).  Returns (batch, x_d, tp_d).
        ``swipe_grid``: build the frame times as swipe() does, arange * 0.005 (world/swipe.py:16-17), instead of
        arange * frame_period / 1000 (world/dio.py:28-29): the two expressions differ in the last bit on some frames."""
        rt = self.rt
        xs = [np.asarray(x, dtype=np.float64) for x in xs]
        lens = [len(x) for x in xs]
        nfs = [_tables.frame_count(n, fs, frame_period) for n in lens]
        batch = rt.make_batch(np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(nfs)]))
        x_d = rt.to_device_concat(xs)
        if swipe_grid:
            _require_swipe_period(frame_period)
            tp_h = np.concatenate([np.arange(0, n) * SWIPE_DT for n in nfs])
        else:
            tp_h = np.concatenate([_tables.frame_times(n, frame_period) for n in nfs])
        tp_d = rt.to_device(tp_h)
        batch.tp_d, batch.tp_host = tp_d, tp_h  # the frame grid belongs to the batch descriptor (no pointer-keyed lookup)
        return batch, x_d, tp_d

    @_on_lane_stream
    def encode_device(self, batch, x_d, tp_d, fs, f0_method='dio', f0_floor=71, f0_ceil=800, channels_in_octave=2,
                      target_fs=4000, frame_period=5, allowed_range=0.1, fft_size=None, is_requiem=False,
                      f0_done=None, check=True, want_ps=False, event_caps=None):
        """world/main.py:106-152 for a resident batch.  tp_d is not modified (a copy is kept in the result).
        ``f0_done``: optional callable invoked once the F0 stage has been enqueued (used to stagger lanes).
        ``check``: True — read the sticky device flags afterwards (synchronises this lane's stream) and raise
        WorldHipError if a kernel reported a condition; False — keep the call asynchronous and check later
        (``WorldBatch.check()``); ``'deferred'`` — no host wait either: the flags are published by a kernel behind this
        call's work (wh_flags_post) and the NEXT deferred-check call (or ``check()``) raises for them — late, never lost:
        the mode for a caller that keeps batches in flight.
        ``want_ps``: keep CheapTrick's complex spectra as ``enc.ps_spectrogram`` (encode()'s 'ps spectrogram').
        ``event_caps`` (Harvest): capacities of its zero-crossing lists, see ``world.harvest.harvest_device``.  With
        ``check=True`` a call whose estimate was exceeded (WH_FLAG_EVENT_OVERFLOW: stretches constant up to rounding, e.g.
        digital silence next to signal) is repeated once with the capacities it counted; an asynchronous call reports
        the condition and the caller repeats it (``event_caps='safe'`` or ``counted_event_caps``)."""
        rt = self.rt
        self._deferred_begin(check, "encode_device")
        if fft_size is not None:
            f0_floor = 3.0 * fs / fft_size
        if f0_method == 'dio':
            f0_d, vuv_d, _, _ = dio_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, channels_in_octave, target_fs,
                                           frame_period, allowed_range)
            f0_d = stonemask_device(rt, batch, x_d, tp_d, f0_d, fs, f0_floor)
        elif f0_method == 'harvest':
            from .harvest import harvest_device
            f0_d, vuv_d = harvest_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period, event_caps=event_caps)
        elif f0_method == 'swipe':
            from .swipe import swipe_device
            # world/main.py:134-135 calls swipe() with its default dt = 5 ms whatever frame_period says, and every
            # later stage runs on swipe's own 5 ms grid: a batch grid of another period cannot reproduce that
            _require_swipe_period(frame_period)
            f0_d, vuv_d = swipe_device(rt, batch, x_d, fs, (f0_floor, f0_ceil), SWIPE_DT, 0.3)
        else:
            raise Exception
        if f0_done is not None:
            f0_done()
        ct_fft = int(fft_size) if fft_size is not None else default_fft_size(fs)
        tp_host = batch.tp_host if getattr(batch, "tp_d", None) is tp_d else None
        timebase = None
        if self.prefetch_timebase and not is_requiem and tp_host is not None:
            timebase = self._prefetch_timebase(batch, tp_d, tp_host, f0_d, vuv_d, fs, ct_fft)
        spec_d, ps_d = cheaptrick_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, ct_fft, want_ps=want_ps)
        if is_requiem:
            ap_d = d4c_requiem_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, 0.85, fft_size)
        else:
            ap_d, _ = d4c_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, 0.85, ct_fft)
        if check == 'deferred':
            rt.post_flags()
        elif check:
            flags = rt.take_flags()
            if flags[_hip.FLAG_EVENT_OVERFLOW] and f0_method == 'harvest' and event_caps is None:
                # more crossings than estimated: everything behind Harvest worked on an unusable contour (whatever else
                # it reported goes with it) — once more, with the capacities this pass counted
                from .harvest import counted_event_caps
                if self._tb_rt is not None:
                    self._tb_rt.take_flags()
                return self.encode_device(batch, x_d, tp_d, fs, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs,
                                          frame_period, allowed_range, fft_size, is_requiem, None, True, want_ps,
                                          counted_event_caps(rt))
            rt.raise_for_flags(flags, "encode_device")
        enc = BatchEncoding(rt, batch, fs, tp_d.clone(), f0_d, vuv_d, spec_d, ap_d, ct_fft, is_requiem, frame_period,
                            tp_host=None if tp_host is None else tp_host.copy(), ps_spectrogram=ps_d)
        if timebase is not None:
            # join: behind CheapTrick and D4C in stream order, so the overlap has already happened — and the fork is
            # closed inside this call (a caller that captures encode_device alone in a graph gets a well-formed one)
            rt.torch.cuda.current_stream(rt.device).wait_event(timebase["done"])
            timebase["stamp"] = enc._stamp()
            if timebase["stamp"] is not None:
                enc._timebase = timebase
        return enc

    def _deferred_begin(self, check, where):
        """check='deferred': raise for what earlier deferred-check calls have published by now (no host wait)."""
        if check != 'deferred':
            return
        flags = self.rt.poll_flags()
        if self._tb_rt is not None:
            flags = [a | b for a, b in zip(flags, self._tb_rt.poll_flags())]
        self.rt.raise_for_flags(flags, where + " (condition reported by an earlier call with check='deferred')")

    @_on_lane_stream
    def refill_from_pinned(self, x_d, x_pin):
        """Overwrite the resident waveform tensor ``x_d`` from the pinned host tensor ``x_pin`` with a KERNEL that reads
        the mapped host memory (wh_copy_mapped), stream-ordered on the lane's stream.  A plain ``copy_`` would queue
        the upload on a DMA engine; when a long download from ``download_async`` sits on the same engine the upload —
        and with it the whole next step — waits for it (observed as 33-36 instead of 24 ms per pipelined step)."""
        rt = self.rt
        assert x_pin.is_pinned() and x_pin.numel() == x_d.numel() and x_pin.dtype == x_d.dtype
        _hip.check(rt.lib.wh_copy_mapped(rt.ctx, rt.stream(), rt.ptr(x_d), _hip._vp(x_pin.data_ptr()),
                                         x_d.numel() * x_d.element_size(), 0))
        return x_d

    def download_async(self, tensors, slot=0, mapped=False):
        """Start copying device ``tensors`` into this object's pinned host buffers of ``slot`` (two slots) on a private
        copy stream, behind everything enqueued so far on the lane's stream, and return ``(host_tensors, event)``:
        the lane can go on with the next batch at once, ``event.synchronize()`` (or ``download_wait``) says when the
        host copies are complete.  A streaming caller alternates the slots: with the results of one step leaving
        over PCIe (its own copy engine) under the upload and kernels of the next, throughput is set by the slower of
        the two instead of their sum (bench.py ``with_transfers_pipelined``: 24 against 35.6 ms for config 2).
        Re-using a slot waits for its previous download first.  The pinned buffers are reallocated only when shapes
        change."""
        torch = self.rt.torch
        st = getattr(self, "_dl", None)
        if st is None:
            st = self._dl = {"stream": torch.cuda.Stream(device=self.rt.device), "pins": [None, None], "done": [None, None]}
        tensors = list(tensors)
        pins = st["pins"][slot]
        if pins is None or len(pins) != len(tensors) or any(
                p.shape != t.shape or p.dtype != t.dtype for p, t in zip(pins, tensors)):
            pins = st["pins"][slot] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in tensors]
        if st["done"][slot] is not None:
            st["done"][slot].synchronize()  # the previous contents of this slot have been handed on
        with self.rt.on_stream():
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.rt.device))
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_event(ready)
            for p, t in zip(pins, tensors):
                if mapped and t.is_contiguous() and (t.numel() * t.element_size()) % 8 == 0:
                    # a few workgroups write the pinned buffer through its device mapping: no DMA queue involved
                    _hip.check(self.rt.lib.wh_copy_mapped(self.rt.ctx, _hip._vp(st["stream"].cuda_stream),
                                                          _hip._vp(p.data_ptr()), self.rt.ptr(t),
                                                          t.numel() * t.element_size(), 16))
                else:
                    p.copy_(t, non_blocking=True)
                t.record_stream(st["stream"])
            done = torch.cuda.Event()
            done.record(st["stream"])
        st["done"][slot] = done
        return pins, done

    @staticmethod
    def download_wait(event):
        event.synchronize()

    def check(self, where="WorldBatch"):
        """Read-and-clear the device condition flags of this lane; raises WorldHipError if any is set."""
        with self.rt.on_stream():
            flags = self.rt.take_flags()
            if self._tb_rt is not None:  # conditions raised by a prefetched time base
                flags = [a | b for a, b in zip(flags, self._tb_rt.take_flags())]
            return self.rt.raise_for_flags(flags, where)

    def encode(self, xs, fs, **kw):
        if kw.get('f0_method') == 'swipe':
            # the reference calls swipe() with its default dt = 5 ms whatever frame_period says and every later stage
            # runs on that grid (world/main.py:134-135): frame_period is ignored, as in World.encode
            kw = dict(kw, frame_period=5)
        batch, x_d, tp_d = self.upload(xs, fs, kw.get('frame_period', 5), swipe_grid=kw.get('f0_method') == 'swipe')
        if kw.get('f0_method') == 'harvest' and kw.get('event_caps') is None:
            from .harvest import flat_samples
            batch.flat_samples = [flat_samples(x) for x in xs]  # (40 us per 10 s: digital silence needs longer crossing lists)
        enc = self.encode_device(batch, x_d, tp_d, fs, **kw)
        if kw.get('f0_method') == 'harvest' and kw.get('check', True) is not True and kw.get('event_caps') is None:
            # an asynchronous Harvest encode: whoever reads its flags can repeat it (settle_encode)
            enc._repeat = (batch, x_d, tp_d, fs, {k: v for k, v in kw.items() if k not in ('check', 'f0_done')})
        return enc

    def settle_encode(self, enc, where="WorldBatch"):
        """``check()`` for a lane whose last work was the asynchronous ``encode`` that returned ``enc`` (the lane's stream has
        been waited for): raises like ``check()`` — except for Harvest's WH_FLAG_EVENT_OVERFLOW (more zero crossings than
        estimated: stretches constant up to rounding, e.g. digital silence next to signal), where the encode is repeated
        with the capacities the first pass counted.  Returns the encoding to use."""
        with self.rt.on_stream():
            flags = self.rt.take_flags()
            tb_flags = self._tb_rt.take_flags() if self._tb_rt is not None else [0] * len(flags)
            rep = getattr(enc, "_repeat", None)
            if rep is not None:
                enc._repeat = None
            if not (flags[_hip.FLAG_EVENT_OVERFLOW] and rep is not None):
                self.rt.raise_for_flags([a | b for a, b in zip(flags, tb_flags)], where)
                return enc
            from .harvest import counted_event_caps
            batch, x_d, tp_d, fs, kw = rep
            caps = counted_event_caps(self.rt)
        return self.encode_device(batch, x_d, tp_d, fs, check=True, event_caps=caps, **kw)

    @_on_lane_stream
    def upload_pcm16(self, pcm_list, fs, frame_period=5):
        """upload() for 16-bit PCM (what scipy.io.wavfile.read returns): the int16 samples cross PCIe (a quarter of
        the float64 bytes) and are scaled on the device like the reference's callers do, x = pcm / (2**15 - 1)
        (example/prosody.py:13).  Returns (batch, x_d, tp_d)."""
        rt = self.rt
        pcm_list = [np.ascontiguousarray(p, dtype=np.int16) for p in pcm_list]
        lens = [len(p) for p in pcm_list]
        nfs = [_tables.frame_count(n, fs, frame_period) for n in lens]
        batch = rt.make_batch(np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(nfs)]))
        pcm_d = rt.torch.from_numpy(np.concatenate(pcm_list)).to(rt.device)
        x_d = rt.empty((int(sum(lens)),))
        _hip.check(rt.lib.wh_pcm16_to_f64(rt.ctx, rt.stream(), rt.ptr(pcm_d), int(sum(lens)), rt.ptr(x_d)))
        tp_h = np.concatenate([_tables.frame_times(n, frame_period) for n in nfs])
        tp_d = rt.to_device(tp_h)
        batch.tp_d, batch.tp_host = tp_d, tp_h
        return batch, x_d, tp_d

    @_on_lane_stream
    def to_pcm16(self, y, y_off):
        """List of per-utterance int16 arrays from decode_device's output, (y * 2**15).astype(int16) evaluated on the
        device (example/prosody.py:57): a quarter of the float64 bytes come back over PCIe."""
        rt = self.rt
        pcm_d = rt.empty((int(y.shape[0]),), dtype=rt.torch.int16)
        _hip.check(rt.lib.wh_f64_to_pcm16(rt.ctx, rt.stream(), rt.ptr(y), int(y.shape[0]), rt.ptr(pcm_d)))
        pcm = pcm_d.cpu().numpy()
        return [pcm[int(y_off[u]):int(y_off[u + 1])].copy() for u in range(len(y_off) - 1)]

    @_on_lane_stream
    def decode_device(self, enc, noise=None, seed=0, pulse_cap=None, seeds=None, cursor=None, check=True):
        """world/main.py:198-214 for a resident encoding.  Returns (y tensor, y_off) — concatenated
        waveforms, peak-normalised per utterance where max|y| > 1.  ``noise``: optional list of per-utterance
        standard-normal arrays (reference-parity mode); default = on-device Philox stream ``seed``.
        Requiem encodings: ``seeds`` = get_seeds_signals(fs) tables (built and cached per fs when omitted),
        ``cursor`` = read position in the circular noise seed at which the first utterance starts (default 0 = a
        fresh reference process); utterances consume the seed like consecutive reference calls.

        Pulse capacity: ``pulse_cap`` slots per utterance; the default max(ny)//8+64 covers a mean f0 below fs/8.
        With ``check`` (default) the sticky device flags are read afterwards (synchronises the stream): an overflow
        of the DEFAULT capacity re-runs the decode with the safe bound ny//2+16 (f0 < fs/2), any other condition —
        or an overflow of an explicit ``pulse_cap`` — raises WorldHipError.  ``check=False`` keeps the call
        asynchronous; call ``WorldBatch.check()`` before trusting the audio.  ``check='deferred'``: as in
        encode_device — published behind the work, raised by the next deferred-check call or ``check()``."""
        rt = self.rt
        self._deferred_begin(check, "decode_device")
        fo = enc.batch.frame_off
        tb = None
        if not enc.is_requiem:
            tb = enc.timebase_for(self, pulse_cap)
        if tb is not None:
            geo = tb["geo"]
        else:
            tp_h = enc.host_times()
            geo = [time_axis_params(tp_h[int(fo[u]):int(fo[u + 1])], enc.fs) for u in range(enc.n_utt)]
        ny = [g[0] for g in geo]
        noise_d = noise_off = None
        if noise is not None and not enc.is_requiem:
            noise_off = np.concatenate([[0], np.cumsum([len(n) for n in noise])])
            noise_d = rt.to_device(np.concatenate(noise))

        def run(cap):
            if enc.is_requiem:
                from .synthesisRequiem import synthesis_requiem_device
                y, y_off = synthesis_requiem_device(rt, enc, ny, geo, seeds=seeds, cursor=cursor, pulse_cap=cap)
            else:
                use_tb = tb is not None and (cap is None or cap == tb["pulse_cap"])
                if use_tb:  # join: the render goes behind the prefetched time base
                    rt.torch.cuda.current_stream(rt.device).wait_event(tb["done"])
                    tb["rt"].timebase_consumed = True  # its conditions are now owed to this caller (_prefetch_timebase)
                y, y_off = synthesis_device(rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, enc.spectrogram,
                                            enc.aperiodicity, enc.fs, enc.fft_size, ny, [g[1] for g in geo],
                                            [g[2] for g in geo], noise_d=noise_d, noise_off=noise_off, seed=seed,
                                            pulse_cap=tb["pulse_cap"] if use_tb else cap,
                                            timebase_rt=tb["rt"] if use_tb else None)
            self._peak_normalise(y, y_off)
            return y, y_off

        y, y_off = run(pulse_cap)
        if check == 'deferred':
            # (an overflow of the default pulse capacity cannot be retried here: it is raised at the next poll — pass
            # pulse_cap=safe_pulse_cap(...) for material whose mean f0 may exceed fs/8)
            rt.post_flags()
            if tb is not None:
                with rt.torch.cuda.stream(tb["rt"].own_stream):
                    tb["rt"].post_flags()
        elif check:
            flags = rt.take_flags()
            if tb is not None:  # conditions raised by the time-base kernels live in that context's flags
                flags = [a | b for a, b in zip(flags, tb["rt"].take_flags())]
            rt.raise_for_flags(flags, "decode_device", allow=() if pulse_cap is not None else (_hip.FLAG_PULSE_OVERFLOW,))
            if flags[_hip.FLAG_PULSE_OVERFLOW]:
                y, y_off = run(safe_pulse_cap(ny))
                rt.check_flags("decode_device")
        return y, y_off

    @_on_lane_stream
    def settle_decode(self, enc, result, noise=None, seed=0, pulse_cap=None, seeds=None, cursor=None, check=None):
        """The ending of ``decode_device(check=True)`` for a decode that was enqueued with ``check=False`` (same keywords)
        once its stream has been waited for: reads the condition flags; an overflow of the DEFAULT pulse capacity
        renders again with the safe one, anything else raises.  Returns the (y, y_off) to use."""
        rt = self.rt
        flags = rt.take_flags()
        if self._tb_rt is not None:
            flags = [a | b for a, b in zip(flags, self._tb_rt.take_flags())]
        rt.raise_for_flags(flags, "decode_device", allow=() if pulse_cap is not None else (_hip.FLAG_PULSE_OVERFLOW,))
        if flags[_hip.FLAG_PULSE_OVERFLOW]:
            y_off = result[1]
            ny = [int(y_off[u + 1] - y_off[u]) for u in range(len(y_off) - 1)]
            return self.decode_device(enc, noise=noise, seed=seed, pulse_cap=safe_pulse_cap(ny), seeds=seeds,
                                      cursor=cursor, check=True)
        return result

    def _peak_normalise(self, y, y_off):
        """y /= max|y| where it exceeds 1 (world/main.py:209-212), per utterance, on the device."""
        import ctypes

        rt = self.rt
        off = np.ascontiguousarray(y_off, dtype=np.int64)
        _hip.check(rt.lib.wh_peak_normalise(rt.ctx, rt.stream(), rt.ptr(y), off.ctypes.data_as(ctypes.c_void_p),
                                            len(off) - 1))


def _check_all(batches, where):
    """WorldBatch.check() of every one of ``batches`` — all of them are read (and cleared) even if one raises: a condition
    left standing in a context would be blamed on the next batch that runs there — then the first error is raised."""
    err = None
    for wb in batches:
        try:
            wb.check(where)
        except _hip.WorldHipError as e:
            err = err or e
    if err is not None:
        raise err


class WorldBatchLanes:
    """The same resident pipeline with the utterances dealt to ``lanes`` independent sub-batches, each on its own
    HIP stream and library context (``_hip.Runtime`` lanes).  Nothing is exchanged between lanes: utterances are
    independent, so the GPU is free to run one lane's chip-filling kernels (D4C,
    CheapTrick, the pulse responses) while another lane sits in its per-utterance serial kernels (the IIR chains,
    the exact phase scan, pulse compaction), which occupy a handful of CUs.  ``lanes=1`` is plain WorldBatch.
    """

    def __init__(self, device_index=None, lanes=2, first_lane=None):
        """``first_lane``: lane id of the first sub-batch (default: 0 for a single lane = torch's current stream, 1.. for
        several).  Two single-lane objects with different ids are two independent pipelines on one GPU — contexts,
        workspaces and streams of their own — e.g. for two whole batches in flight (bench.py --in-flight)."""
        if first_lane is None:
            first_lane = 1 if lanes > 1 else 0
        self.lanes = [WorldBatch(device_index, lane=first_lane + i) for i in range(lanes)]
        self.resident = None

    @staticmethod
    def split(lengths, lanes):
        """Contiguous utterance ranges per lane, balanced by samples (the rule ranks are sharded by)."""
        from .distributed import shard_ranges
        return shard_ranges(lengths, lanes)

    def upload(self, xs, fs, frame_period=5):
        parts = self.split([len(x) for x in xs], len(self.lanes))
        self.resident = [wb.upload(xs[a:b], fs, frame_period) if b > a else None
                         for wb, (a, b) in zip(self.lanes, parts)]
        return self.resident

    @property
    def total_frames(self):
        return sum(r[0].total_frames for r in self.resident if r is not None)

    def encode_device(self, fs, stagger=True, **kw):
        """One BatchEncoding per lane (None for an empty lane); launches are asynchronous.

        ``stagger``: lane i starts once lane i-1 has finished its F0 stage (a stream-to-stream event wait, no host
        sync).  Lanes that start together stay in lockstep — all in their serial kernels at once, then all
        competing for the CUs at once; the offset is what puts one lane's serial stretch under another lane's
        chip-filling kernels, and it persists from step to step."""
        torch = self.lanes[0].rt.torch
        out, prev = [], None
        for wb, r in zip(self.lanes, self.resident):
            if r is None:
                out.append(None)
                continue
            if prev is not None and wb.rt.own_stream is not None:
                wb.rt.own_stream.wait_event(prev)
            ev = torch.cuda.Event() if stagger and wb.rt.own_stream is not None else None
            kw.setdefault("check", False)  # lanes stay asynchronous: flags are read in synchronize()
            out.append(wb.encode_device(r[0], r[1], r[2], fs, f0_done=(ev.record if ev is not None else None), **kw))
            prev = ev
        return out

    def decode_device(self, encs, **kw):
        """[(y, y_off) per lane]; asynchronous (``check=False`` unless asked otherwise): call synchronize()."""
        kw.setdefault("check", False)
        return [wb.decode_device(e, **kw) if e is not None else None for wb, e in zip(self.lanes, encs)]

    def synchronize(self, check=True):
        """Wait for every lane; with ``check`` raise WorldHipError if a kernel of any lane reported a condition."""
        for wb in self.lanes:
            if wb.rt.own_stream is not None:
                wb.rt.own_stream.synchronize()
            else:
                wb.rt.torch.cuda.current_stream(wb.rt.device).synchronize()
        if check:
            _check_all(self.lanes, "WorldBatchLanes")


class WorldBatchPipeline:
    """``depth`` independent WorldBatch pipelines on one GPU — contexts, workspace arenas and HIP streams of their own —
    handed out round-robin: a caller that processes a stream of batches runs batch k on ``pipeline.next()``.  Nothing is
    shared between two batches in flight and nothing is skipped; what two in flight buy is that the serial,
    latency-bound head of one batch's encode (decimation IIRs, contour tracking: a few dozen workgroups) runs under the
    chip-filling kernels at the tail of the batch before instead of on an idle chip (bench.py --in-flight: config 2
    10.09 -> 9.78 ms per 64 x 10 s step, config 4 13.5 -> 12.6).  Calls are asynchronous by default (``check=False``):
    ``synchronize()`` waits for every pipeline and raises for the conditions their kernels reported.  Memory: every
    pipeline keeps an arena sized for its largest batch (1024 x 10 s of Harvest: ~105 GB — one in flight at that size)."""

    def __init__(self, device_index=None, depth=2, prefetch_timebase=True, first_lane=1):
        """``first_lane``: lane id of the first pipeline (lanes are contexts + streams, `_hip.Runtime`): two pipeline
        objects with the same lane ids SHARE contexts, flags and streams — the facade (world/main.py) keeps ids of its
        own so that its flag reads never consume a condition of the caller's own pipelines in flight."""
        self.pipes = [WorldBatch(device_index, lane=first_lane + d, prefetch_timebase=prefetch_timebase)
                      for d in range(max(1, int(depth)))]
        self._k = 0

    def next(self):
        wb = self.pipes[self._k % len(self.pipes)]
        self._k += 1
        return wb

    def encode_decode(self, xs, fs, decode_kw=None, **encode_kw):
        """One batch through the next pipeline: (encoding, y, y_off), enqueued on that pipeline's stream."""
        wb = self.next()
        encode_kw.setdefault("check", False)
        enc = wb.encode(xs, fs, **encode_kw)
        kw = dict(decode_kw or {})
        kw.setdefault("check", False)
        y, y_off = wb.decode_device(enc, **kw)
        return enc, y, y_off

    def synchronize(self, check=True):
        for wb in self.pipes:
            wb.rt.own_stream.synchronize()
        if check:
            _check_all(self.pipes, "WorldBatchPipeline")

