"""Batched, device-resident WORLD pipeline (not in the reference API: it is the harness the benchmark
configs need — SURVEY.md §8(b) "World.encode_batch/decode_batch").

Utterances are concatenated into one waveform tensor; every stage runs ONE launch per kernel over all
frames x utterances.  Results stay in HBM as torch tensors (frame-major spectrogram / aperiodicity);
`BatchEncoding.to_dicts()` materialises reference-layout NumPy dicts on demand.
"""
import numpy as np

from . import _hip, _tables
from .cheaptrick import cheaptrick_device, default_fft_size
from .d4c import d4c_device
from .d4cRequiem import d4c_requiem_device
from .dio import dio_device
from .stonemask import stonemask_device
from .synthesis import synthesis_device, time_axis_params


class BatchEncoding:
    def __init__(self, rt, batch, fs, tp, f0, vuv, spectrogram, aperiodicity, fft_size, is_requiem, frame_period,
                 tp_host=None):
        self.rt, self.batch, self.fs = rt, batch, fs
        self.tp_host = tp_host  # host copy of the frame times (kept in step with scale_duration): no D2H in decode
        self.temporal_positions, self.f0, self.vuv = tp, f0, vuv
        self.spectrogram, self.aperiodicity = spectrogram, aperiodicity
        self.fft_size, self.is_requiem, self.frame_period = fft_size, is_requiem, frame_period

    @property
    def n_utt(self):
        return self.batch.n_utt

    def scale_pitch(self, factor):
        """world/main.py:154-162, on the device."""
        self.f0 *= factor
        return self

    def scale_duration(self, factor):
        """world/main.py:170-178, on the device."""
        self.temporal_positions *= factor
        if self.tp_host is not None:
            self.tp_host = self.tp_host * factor
        return self

    def to_dicts(self):
        """List of per-utterance dicts with the reference's keys and (bins, frames) layouts."""
        fo = self.batch.frame_off
        tp, f0, vuv = (t.cpu().numpy() for t in (self.temporal_positions, self.f0, self.vuv))
        sp = self.spectrogram.cpu().numpy()
        ap = self.aperiodicity.cpu().numpy()
        out = []
        for u in range(self.n_utt):
            s = slice(int(fo[u]), int(fo[u + 1]))
            out.append({'temporal_positions': tp[s].copy(), 'vuv': vuv[s].copy(), 'fs': self.fs, 'f0': f0[s].copy(),
                        'aperiodicity': np.ascontiguousarray(ap[s].T), 'spectrogram': np.ascontiguousarray(sp[s].T),
                        'is_requiem': self.is_requiem})
        return out


def _on_lane_stream(fn):
    """Run a WorldBatch method with its lane's HIP stream as torch's current stream."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with self.rt.on_stream():
            return fn(self, *a, **kw)
    return wrapped


class WorldBatch:
    def __init__(self, device_index=None, lane=0):
        self.rt = _hip.Runtime.get(device_index, lane)

    @_on_lane_stream
    def upload(self, xs, fs, frame_period=5):
        """Concatenate and upload a list of 1-D waveforms (or a 2-D array).  Returns (batch, x_d, tp_d)."""
        rt = self.rt
        xs = [np.asarray(x, dtype=np.float64) for x in xs]
        lens = [len(x) for x in xs]
        nfs = [_tables.frame_count(n, fs, frame_period) for n in lens]
        batch = rt.make_batch(np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(nfs)]))
        x_d = rt.to_device(np.concatenate(xs))
        tp_h = np.concatenate([_tables.frame_times(n, frame_period) for n in nfs])
        tp_d = rt.to_device(tp_h)
        self._tp_host = {tp_d.data_ptr(): tp_h}
        return batch, x_d, tp_d

    @_on_lane_stream
    def encode_device(self, batch, x_d, tp_d, fs, f0_method='dio', f0_floor=71, f0_ceil=800, channels_in_octave=2,
                      target_fs=4000, frame_period=5, allowed_range=0.1, fft_size=None, is_requiem=False,
                      f0_done=None):
        """world/main.py:106-152 for a resident batch.  tp_d is not modified (a copy is kept in the result).
        ``f0_done``: optional callable invoked once the F0 stage has been enqueued (used to stagger lanes)."""
        rt = self.rt
        if fft_size is not None:
            f0_floor = 3.0 * fs / fft_size
        if f0_method == 'dio':
            f0_d, vuv_d, _, _ = dio_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, channels_in_octave, target_fs,
                                           frame_period, allowed_range)
            f0_d = stonemask_device(rt, batch, x_d, tp_d, f0_d, fs, f0_floor)
        elif f0_method == 'harvest':
            from .harvest import harvest_device
            f0_d, vuv_d = harvest_device(rt, batch, x_d, tp_d, fs, f0_floor, f0_ceil, frame_period)
        else:
            raise Exception
        if f0_done is not None:
            f0_done()
        ct_fft = int(fft_size) if fft_size is not None else default_fft_size(fs)
        spec_d, _ = cheaptrick_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, ct_fft)
        if is_requiem:
            ap_d = d4c_requiem_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, 0.85, fft_size)
        else:
            ap_d, _ = d4c_device(rt, batch, x_d, tp_d, f0_d, vuv_d, fs, 0.85, ct_fft)
        tp_host = getattr(self, "_tp_host", {}).get(tp_d.data_ptr())
        return BatchEncoding(rt, batch, fs, tp_d.clone(), f0_d, vuv_d, spec_d, ap_d, ct_fft, is_requiem, frame_period,
                             tp_host=None if tp_host is None else tp_host.copy())

    def encode(self, xs, fs, **kw):
        batch, x_d, tp_d = self.upload(xs, fs, kw.get('frame_period', 5))
        return self.encode_device(batch, x_d, tp_d, fs, **kw)

    @_on_lane_stream
    def decode_device(self, enc, noise=None, seed=0, pulse_cap=None, seeds=None):
        """world/main.py:198-214 for a resident encoding.  Returns (y tensor, y_off) — concatenated
        waveforms, peak-normalised per utterance where max|y| > 1.  ``noise``: optional list of per-utterance
        standard-normal arrays (reference-parity mode); default = on-device Philox stream ``seed``."""
        rt = self.rt
        fo = enc.batch.frame_off
        tp_h = enc.tp_host if enc.tp_host is not None else enc.temporal_positions.cpu().numpy()
        geo = [time_axis_params(tp_h[int(fo[u]):int(fo[u + 1])], enc.fs) for u in range(enc.n_utt)]
        ny = [g[0] for g in geo]
        if enc.is_requiem:
            from .synthesisRequiem import synthesis_requiem_device
            y, y_off = synthesis_requiem_device(rt, enc, ny, geo, seeds=seeds)
        else:
            noise_d = noise_off = None
            if noise is not None:
                noise_off = np.concatenate([[0], np.cumsum([len(n) for n in noise])])
                noise_d = rt.to_device(np.concatenate(noise))
            y, y_off = synthesis_device(rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, enc.spectrogram,
                                        enc.aperiodicity, enc.fs, enc.fft_size, ny, [g[1] for g in geo],
                                        [g[2] for g in geo], noise_d=noise_d, noise_off=noise_off, seed=seed,
                                        pulse_cap=pulse_cap)
        self._peak_normalise(y, y_off)
        return y, y_off

    def _peak_normalise(self, y, y_off):
        """y /= max|y| where it exceeds 1 (world/main.py:209-212), per utterance, on the device."""
        import ctypes

        rt = self.rt
        off = np.ascontiguousarray(y_off, dtype=np.int64)
        _hip.check(rt.lib.wh_peak_normalise(rt.ctx, rt.stream(), rt.ptr(y), off.ctypes.data_as(ctypes.c_void_p),
                                            len(off) - 1))


class WorldBatchLanes:
    """The same resident pipeline with the utterances dealt to ``lanes`` independent sub-batches, each on its own
    HIP stream and library context (``_hip.Runtime`` lanes).  Nothing is exchanged between lanes: utterances are
    independent, so the GPU is free to run one lane's chip-filling kernels (D4C,
    CheapTrick, the pulse responses) while another lane sits in its per-utterance serial kernels (the IIR chains,
    the exact phase scan, pulse compaction), which occupy a handful of CUs.  ``lanes=1`` is plain WorldBatch.
    """

    def __init__(self, device_index=None, lanes=2):
        self.lanes = [WorldBatch(device_index, lane=(i + 1 if lanes > 1 else 0)) for i in range(lanes)]
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
            out.append(wb.encode_device(r[0], r[1], r[2], fs, f0_done=(ev.record if ev is not None else None), **kw))
            prev = ev
        return out

    def decode_device(self, encs, **kw):
        """[(y, y_off) per lane]."""
        return [wb.decode_device(e, **kw) if e is not None else None for wb, e in zip(self.lanes, encs)]

    def synchronize(self):
        for wb in self.lanes:
            if wb.rt.own_stream is not None:
                wb.rt.own_stream.synchronize()
            else:
                wb.rt.torch.cuda.current_stream(wb.rt.device).synchronize()
