"""`World` facade: the class, method names, argument order / defaults and result-dict keys of the reference's
world/main.py:26-214, with every stage running on the MI355X through libworld_hip.so.

The analysis/synthesis path and the spectral feature heads that follow encode() (mel filterbank energies, mel-cepstrum
and its inverse, context stacking: world/features.py) are provided; the reference's plotting and Keras/VAE glue
(draw, encode_vae) are not part of this build.
"""
import logging
import os

import numpy as np

from . import _hip
from . import cheaptrick as _ct
from . import d4c as _d4c
from . import d4cRequiem as _d4cr
from . import features as _feat
from . import dio as _dio
from . import get_seeds_signals as _seeds
from . import harvest as _hv
from . import stonemask as _sm
from . import swipe as _swipe
from . import synthesis as _syn
from . import synthesisRequiem as _synr

_SOURCE_KEYS = ('temporal_positions', 'vuv', 'f0')


def _pick(d, keys):
    return {k: d[k] for k in keys}


def _estimate_source(fs, x, method, floor, ceil, channels, target_fs, period, allowed_range):
    """F0 stage shared by get_f0 / get_spectrum / encode (world/main.py:36-46, 62-72, 121-135).  allowed_range is None
    for the two getters, which call dio() positionally without it."""
    if np.ndim(x) != 1:
        raise ValueError("the waveform must be 1-D (one channel), got shape %s" % (np.shape(x),))
    if method == 'dio':
        extra = {} if allowed_range is None else {'allowed_range': allowed_range}
        source = _dio.dio(x, fs, floor, ceil, channels, target_fs, period, **extra)
        source['f0'] = _sm.stonemask(x, fs, source['temporal_positions'], source['f0'])
        return source
    if method == 'harvest':
        return _hv.harvest(x, fs, f0_floor=floor, f0_ceil=ceil, frame_period=period)
    if method == 'swipe':
        return _swipe.swipe(fs, x, plim=[floor, ceil], sTHR=0.3)
    raise Exception  # the reference raises a bare Exception for an unknown method


def _aperiodicity(x, fs, source, fft_size, is_requiem):
    if is_requiem:
        return _d4cr.d4cRequiem(x, fs, source, fft_size=fft_size)
    return _d4c.d4c(x, fs, source, fft_size_for_spectrum=fft_size)


# World.encode_batch / decode_batch cut a batch of at least this many bytes of waveform (audio to be rendered) into two
# parts that run on two pipelines, so that one part's PCIe transfer lies under the other's kernels
FACADE_SPLIT_BYTES = int(os.environ.get("WH_FACADE_SPLIT_BYTES", str(16 << 20)))
FACADE_LANE = int(os.environ.get("WH_FACADE_LANE", "2001"))  # the pipelines the parts run on: lane ids (contexts, flags, streams) no public class hands out, so the
# facade's flag reads never consume — or raise for — a condition of the caller's own WorldBatchPipeline / WorldBatchLanes
# work in flight
# (Measured and not kept: the later part's stream at high priority, and its decode time base computed ahead under the
# earlier part's responses — 23.0 ... 24.6 ms against 23.4 for the resynthesis flow of 64 x 10 s, within the noise.)


def _decode_groups(dats, kw):
    """Consecutive [start, end) parts of a decode_batch list: one part for small batches, for Requiem (its noise cursor
    runs through the utterances in order) and for a caller that steers the checks itself; otherwise two — cut where the
    list passes from one resident encoding to the next if it does so exactly once (the two halves of encode_batch: each
    keeps its tensors as they are), else balanced by frames."""
    from .distributed import shard_ranges

    n = len(dats)
    if n < 2 or bool(dats[0]['is_requiem']) or any(k in kw for k in ('seeds', 'cursor', 'check')):
        return [(0, n)]
    frames = [len(d['f0']) for d in dats]
    seconds = sum(float(d['temporal_positions'][-1]) for d in dats if len(d['temporal_positions']))
    if 8 * seconds * dats[0]['fs'] < FACADE_SPLIT_BYTES:  # the audio to be rendered
        return [(0, n)]
    encs = [getattr(d, '_enc', None) for d in dats]
    cuts = [i for i in range(1, n) if encs[i] is not encs[i - 1]]
    if len(cuts) == 1:
        return [(0, cuts[0]), (cuts[0], n)]
    return [(a, b) for a, b in shard_ranges(frames, 2) if b > a]


def _hand_out(dats, block, y_off, copy_out):
    """d['out'] for every dict from the batch's host block (page-locked): arrays of their own in pageable memory — the
    copies run on the staging pool's threads, NumPy releases the GIL for them — or, ``copy_out=False``, views."""
    cuts = [(int(y_off[u]), int(y_off[u + 1])) for u in range(len(dats))]
    if not copy_out:
        for d, (a, b) in zip(dats, cuts):
            d['out'] = block[a:b]
        return
    if block.nbytes < (1 << 22) or len(dats) < 4:
        outs = [np.array(block[a:b]) for a, b in cuts]
    else:
        # (allocating the arrays on this thread and only filling them in the pool was measured and is worse: 49 against
        # 28 - 39 ms for the 164 MB of a 64 x 20 s batch; the views of copy_out=False: 23 ms)
        with _hip._copy_pool() as pool:
            outs = list(pool.map(lambda ab: np.array(block[ab[0]:ab[1]]), cuts))
    for d, y in zip(dats, outs):
        d['out'] = y


def _check_decodable(dats):
    """What decode() / decode_batch() are handed: the per-frame arrays of a dict must agree in length and the dense tensors
    be (bins, frames) over those frames; the dicts of one batch must share rate, synthesis path and bin count.  The
    reference fails somewhere inside NumPy on such input (an interp1d shape error, a broadcast error); here the kernels
    index by the batch's frame count, so a short array would be read past its end — refused up front instead.  Dense values
    still resident in HBM (lazy dicts) are what encode made: not looked at, not materialised."""
    fs0 = req0 = k0 = None
    for n, d in enumerate(dats):
        tp, f0, vuv = (np.asarray(d[k]) for k in ('temporal_positions', 'f0', 'vuv'))
        if not (tp.ndim == f0.ndim == vuv.ndim == 1 and len(tp) == len(f0) == len(vuv)):
            raise ValueError("dict %d: temporal_positions / f0 / vuv must be 1-D and of one length (%s, %s, %s)"
                             % (n, tp.shape, f0.shape, vuv.shape))
        if len(tp) < 2:
            raise ValueError("dict %d: fewer than 2 frames" % n)
        for key in ('spectrogram', 'aperiodicity'):
            v = dict.get(d, key)
            if isinstance(v, np.ndarray) and (v.ndim != 2 or v.shape[1] != len(tp)):
                raise ValueError("dict %d: '%s' must be (bins, %d frames), got %s" % (n, key, len(tp), v.shape))
        sp = dict.get(d, 'spectrogram')
        k = sp.shape[0] if isinstance(sp, np.ndarray) else None
        if fs0 is None:
            fs0, req0 = d['fs'], bool(d['is_requiem'])
        elif d['fs'] != fs0 or bool(d['is_requiem']) != req0:
            raise ValueError("dict %d: the dicts of one decode_batch must share fs and is_requiem (%s / %s against %s / %s)"
                             % (n, d['fs'], bool(d['is_requiem']), fs0, req0))
        if k is not None:
            if k0 is None:
                k0 = k
            elif k != k0:
                raise ValueError("dict %d: spectrogram of %d bins in a batch of %d" % (n, k, k0))


class World(object):
    # ---- analysis -----------------------------------------------------------------------------------------
    def get_f0(self, fs, x, f0_method='harvest', f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000,
               frame_period=5):
        """(temporal_positions, f0, vuv) — world/main.py:27-49."""
        s = _estimate_source(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, None)
        return tuple(s[k] for k in ('temporal_positions', 'f0', 'vuv'))

    def get_spectrum(self, fs, x, f0_method='harvest', f0_floor=71, f0_ceil=800, channels_in_octave=2,
                     target_fs=4000, frame_period=5, fft_size=None):
        """world/main.py:51-79."""
        s = _estimate_source(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, None)
        env = _ct.cheaptrick(x, fs, s, fft_size=fft_size)
        out = _pick(s, ('f0', 'temporal_positions'))
        out['fs'] = fs
        out.update(_pick(env, ('ps spectrogram', 'spectrogram')))
        return out

    def encode_w_gvn_f0(self, fs, x, source, fft_size=None, is_requiem=False):
        """world/main.py:81-104, quirks included: fft_size=None raises TypeError at the assert and is_requiem=True
        raises KeyError('coarse_ap')."""
        assert np.all(source['f0'] >= 3 * fs / fft_size)
        env = _ct.cheaptrick(x, fs, source, fft_size=fft_size)
        source = _aperiodicity(x, fs, source, fft_size, is_requiem)
        out = _pick(source, _SOURCE_KEYS)
        out['fs'] = fs
        out['spectrogram'] = env['spectrogram']
        out.update(_pick(source, ('aperiodicity', 'coarse_ap')))
        out['is_requiem'] = is_requiem
        return out

    def encode(self, fs, x, f0_method='harvest', f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000,
               frame_period=5, allowed_range=0.1, fft_size=None, is_requiem=False):
        """world/main.py:106-152."""
        if fft_size != None:  # noqa: E711 — the reference's own test: 0 also counts as "given"
            f0_floor = 3.0 * fs / fft_size
        source = _estimate_source(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period,
                                  allowed_range)
        env = _ct.cheaptrick(x, fs, source, fft_size=fft_size)
        source = _aperiodicity(x, fs, source, fft_size, is_requiem)
        out = _pick(source, ('temporal_positions', 'vuv'))
        out['fs'] = env['fs']
        out.update(_pick(source, ('f0', 'aperiodicity')))
        out.update(_pick(env, ('ps spectrogram', 'spectrogram')))
        out['is_requiem'] = is_requiem
        return out

    # ---- batched convenience (not in the reference; SURVEY.md §8(b): "World.encode_batch/decode_batch") ----------
    @_hip.serialised
    def encode_batch(self, fs, xs, f0_method='harvest', f0_floor=71, f0_ceil=800, channels_in_octave=2, target_fs=4000,
                     frame_period=5, allowed_range=0.1, fft_size=None, is_requiem=False, want_ps=False, devices=None):
        """encode() — same arguments, same defaults (Harvest) — for a list of utterances in one pass per kernel.
        ``devices``: a list of GPU indices — the batch is cut into contiguous ranges balanced by samples and every range
        runs on its device from a host thread of its own (``world.pool.WorldBatchPool``: one process, no
        torch.distributed); the dicts are the single-device ones, bit for bit.
        Each dict has encode()'s keys; 'ps spectrogram' (the complex pitch-synchronous spectra of world/main.py:149,
        fft_size x frames x 16 B per utterance) only with ``want_ps=True``.  Under an initialised
        torch.distributed process group (one process per GPU) the batch is sharded by utterance over the ranks and the
        list holds only this rank's utterances; use ``world.distributed.ShardedWorldBatch`` directly to keep results
        on the device.

        The dicts are ``world.batch.EncodingDict``s: real dicts whose dense values ('spectrogram', 'aperiodicity',
        'ps spectrogram') are downloaded when first read; ``decode_batch`` takes whatever was never read straight from
        HBM.  ``encode_batch -> scale_pitch -> scale_duration -> decode_batch`` moves no dense tensor over PCIe."""
        from .batch import WorldBatchPipeline
        from .distributed import ShardedWorldBatch, shard_ranges

        kw = dict(f0_method=f0_method, f0_floor=f0_floor, f0_ceil=f0_ceil, channels_in_octave=channels_in_octave,
                  target_fs=target_fs, frame_period=frame_period, allowed_range=allowed_range, fft_size=fft_size,
                  is_requiem=is_requiem, want_ps=want_ps)
        if devices is not None:
            from .pool import WorldBatchPool
            dats = WorldBatchPool.shared(devices).encode(list(xs), fs, **kw).to_dicts(want_ps=want_ps, lazy=True)
            for d in dats:
                d['_batch_range'] = (0, len(dats))
            return dats
        sb = ShardedWorldBatch()
        lo, hi = sb.shard([len(x) for x in xs])
        mine = list(xs[lo:hi])
        if len(mine) >= 2 and 8 * sum(len(x) for x in mine) >= FACADE_SPLIT_BYTES:
            # Two halves on two pipelines (contexts and streams of their own): the second half's waveforms cross PCIe
            # under the first half's kernels.  Utterances are independent and every stage numbers its work per
            # utterance, so the dicts are the ones the single batch gives, bit for bit.
            pipe = WorldBatchPipeline(sb.backend.rt.index, depth=2, prefetch_timebase=False, first_lane=FACADE_LANE)
            parts = [(a, b) for a, b in shard_ranges([len(x) for x in mine], 2) if b > a]
            wbs = [pipe.next() for _ in parts]
            encs = [wb.encode(mine[a:b], fs, check=False, **kw) for wb, (a, b) in zip(wbs, parts)]
            pipe.synchronize(check=False)
            err = None
            for i, wb in enumerate(wbs):  # every part's flags are read (and a Harvest part that needs it repeated), then the first error raised
                try:
                    encs[i] = wb.settle_encode(encs[i], "World.encode_batch")
                except _hip.WorldHipError as e:
                    err = err or e
            if err is not None:
                raise err
            dats = [d for e in encs for d in e.to_dicts(want_ps=want_ps, lazy=True)]
        else:
            enc = sb.encode(xs, fs, **kw)
            dats = enc.to_dicts(want_ps=want_ps, lazy=True) if enc is not None else []
        for d in dats:
            d['_batch_range'] = sb.range
        return dats

    @_hip.serialised
    def decode_batch(self, dats, devices=None, copy_out=True, **kw):
        """decode() for a list of encode()/encode_batch() dicts that share fs / is_requiem / fft size: one batched
        synthesis; adds 'out' to every dict (peak-normalised like decode()) and returns the list.
        ``devices``: as in encode_batch — one host thread per listed GPU, each decoding a contiguous range (dicts that
        encode_batch(devices=...) made go back to the device that holds them); same samples as the single batch.
        ``copy_out`` (default): every 'out' is an array of its own in pageable memory, as decode() returns it;
        False: views into ONE page-locked block per batch (part) — no host copy, but a caller that keeps a single
        utterance keeps the whole batch's block locked.
        Randomness differs from decode() unless asked otherwise: the noise comes from the device Philox stream
        (``seed=``) and Requiem batches use device-built seed tables with the noise cursor restarted at 0 on every
        call, whereas decode() draws from NumPy's global stream / get_seeds_signals() and keeps
        ``synthesisRequiem.generate_noise.current_index`` across calls.  For sample-comparable output pass
        ``noise=[randn arrays]`` (pulse-wise) or ``seeds=get_seeds_signals(fs), cursor=...`` (Requiem): keywords of
        ``WorldBatch.decode_device``."""
        from .batch import BatchEncoding, WorldBatch

        if not dats:
            return dats
        _check_decodable(dats)
        if devices is not None:
            from .pool import WorldBatchPool
            for d, y in zip(dats, WorldBatchPool.shared(devices).decode_dicts(dats, **kw)):
                d['out'] = y
            return dats
        groups = _decode_groups(dats, kw)
        if len(groups) > 1:
            return self._decode_batch_parts(dats, groups, kw, copy_out)
        wb = WorldBatch()
        enc = BatchEncoding.from_dicts(wb.rt, dats)
        y, y_off = wb.decode_device(enc, **kw)
        with wb.rt.on_stream():
            y = wb.rt.to_host(y)  # one pinned block
        _hand_out(dats, y, y_off, copy_out)
        return dats

    def _decode_batch_parts(self, dats, groups, kw, copy_out=True):
        """decode_batch with the batch cut into consecutive parts, each on a pipeline of its own (context, stream): part
        g + 1 renders behind part g's kernels, so part g's audio crosses PCIe under them.  Same samples as the single
        batch: the overlap-add runs are numbered per utterance and the Philox stream of utterance u is re-keyed to its
        index in the whole list (synthesis.philox_seed_for_offset)."""
        from .batch import BatchEncoding, WorldBatch

        noise, seed = kw.get('noise'), kw.get('seed', 0)
        index = WorldBatch().rt.index
        pend, prev, wb = [], None, None
        try:
            for g, (a, b) in enumerate(groups):
                wb = WorldBatch(index, lane=FACADE_LANE + g, prefetch_timebase=False)
                rt, torch = wb.rt, wb.rt.torch
                enc = BatchEncoding.from_dicts(rt, dats[a:b])
                kw_g = dict(kw, seed=_syn.philox_seed_for_offset(seed, a), check=False)
                if noise is not None:
                    kw_g['noise'] = noise[a:b]
                if prev is not None:
                    rt.own_stream.wait_event(prev)
                y, y_off = wb.decode_device(enc, **kw_g)
                with rt.on_stream():
                    prev = torch.cuda.Event()
                    prev.record(torch.cuda.current_stream(rt.device))
                    host = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
                    nbytes = y.numel() * y.element_size()
                    if g + 1 < len(groups) and y.is_contiguous() and nbytes % 8 == 0:
                        # a few workgroups write the pinned block through its device mapping.  (The runtime's own
                        # D2H copy is a chip-wide kernel whose waves sit on PCIe: the next part's kernels made no
                        # progress under it — rocprofv3 trace: its first 0.05 ms kernel ended when the 1.5 ms copy did.)
                        _hip.check(rt.lib.wh_copy_mapped(rt.ctx, rt.stream(), _hip._vp(host.data_ptr()), rt.ptr(y),
                                                         nbytes, 16))
                    else:
                        host.copy_(y, non_blocking=True)
                pend.append((wb, enc, kw_g, y, y_off, host))
        except BaseException:
            # nothing of an abandoned batch is left standing in the pipelines' contexts: the parts in flight AND the part
            # whose enqueueing failed (its stream may hold half a decode, its flags a condition of it)
            for w in {id(w): w for w in [p[0] for p in pend] + ([wb] if wb is not None else [])}.values():
                try:
                    w.rt.own_stream.synchronize()
                    w.rt.take_flags()
                except _hip.WorldHipError:
                    pass
            raise
        for wb, *_ in pend:
            wb.rt.own_stream.synchronize()
        settled, err = [], None
        for wb, enc, kw_g, y, y_off, host in pend:  # (every part's conditions are read, whichever raises)
            try:
                settled.append(wb.settle_decode(enc, (y, y_off), **kw_g))
            except _hip.WorldHipError as e:
                settled.append(None)
                err = err or e
        if err is not None:
            raise err
        for (a, b), (wb, enc, kw_g, y, y_off, host), (y2, y_off2) in zip(groups, pend, settled):
            out = host.numpy()
            if y2 is not y:  # rendered again with the safe pulse capacity
                with wb.rt.on_stream():
                    out = wb.rt.to_host(y2)
            _hand_out(dats[a:b], out, y_off2, copy_out)
        return dats

    # ---- modification (all in place on the dict, like the reference) ------------------------------------------
    def scale_pitch(self, dat, factor):
        """world/main.py:154-162."""
        dat['f0'] *= factor
        return dat

    def set_pitch(self, dat, time, value):
        raise NotImplementedError  # world/main.py:164-165

    def scale_duration(self, dat, factor):
        """world/main.py:170-178."""
        dat['temporal_positions'] *= factor
        return dat

    def modify_duration(self, dat, from_time, to_time):
        """Piecewise-linear time map anchored at 0 and at the last frame; returns None (world/main.py:180-189)."""
        tp = dat['temporal_positions']
        last = tp[-1]
        # the reference's checks, written as it writes them (np.all(...) > 0, i.e. "not all steps are zero")
        for axis in (from_time, to_time):
            assert np.all(np.diff(axis)) > 0
        assert from_time[0] > 0 and from_time[-1] < last
        if to_time[-1] == -1:
            to_time[-1] = last
        dat['temporal_positions'] = np.interp(tp, np.r_[0, from_time, last], to_time)

    def warp_spectrum(self, dat, factor):
        """Frequency warping of every frame by f -> f**factor on the normalised axis (world/main.py:191-196)."""
        spec = dat['spectrogram']
        axis = np.arange(spec.shape[0]) / spec.shape[0]
        warped = np.stack([np.interp(axis ** factor, axis, column) for column in spec.T], axis=1)
        spec[:] = warped
        return dat

    # ---- spectral feature heads (world/main.py:257-365; world/features.py) ----------------------------------------
    def hz2mel(self, hz):
        return _feat.hz2mel(hz)

    def mel2hz(self, mel):
        return _feat.mel2hz(mel)

    def get_filterbanks(self, nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
        return _feat.get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq)

    def encode_lfbank(self, spec, prefac=0.97, fs=16000, nfilt=32, lowfreq=0, highfreq=None):
        return _feat.encode_lfbank(spec, prefac, fs, nfilt, lowfreq, highfreq)

    def encode_mcep(self, spec, n0=12, fs=16000, lowhz=0, highhz=8000):
        return _feat.encode_mcep(spec, n0, fs, lowhz, highhz)

    def decode_mcep(self, cepstrum, fft_size):
        return _feat.decode_mcep(cepstrum, fft_size)

    def get_context(self, X, w=5):
        return _feat.get_context(X, w)

    # ---- synthesis ----------------------------------------------------------------------------------------------
    def decode(self, dat):
        """world/main.py:198-214: pulse-wise or Requiem synthesis, then peak normalisation above 1."""
        _check_decodable([dat])
        if dat['is_requiem']:
            y = _synr.synthesisRequiem(dat, dat, _seeds.get_seeds_signals(dat['fs']))
        else:
            y = _syn.synthesis(dat, dat)
        peak = np.max(np.abs(y))
        if peak > 1.0:
            logging.info('rescaling waveform')
            y /= peak
        dat['out'] = y
        return dat
