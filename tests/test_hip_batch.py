"""GPU: the batched device-resident pipeline equals the per-utterance drop-in API, and matches the oracle
end to end (BASELINE config 2 shape at reduced size)."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def test_batch_equals_single_and_oracle():
    from oracle import api as oapi
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.cheaptrick import cheaptrick
    from world.d4c import d4c
    from world.dio import dio
    from world.stonemask import stonemask

    fs = 16000
    xs = [synth_utterance(20 + i, fs, 0.8 + 0.2 * i) for i in range(3)]  # ragged
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method='dio')
    dicts = enc.to_dicts()
    for u, x in enumerate(xs):
        src = dio(x, fs)
        src['f0'] = stonemask(x, fs, src['temporal_positions'], src['f0'])
        filt = cheaptrick(x, fs, src)
        src = d4c(x, fs, src)
        d = dicts[u]
        assert np.array_equal(d['f0'], src['f0'])
        assert np.array_equal(d['vuv'], src['vuv'])
        assert np.array_equal(d['spectrogram'], filt['spectrogram'])
        assert np.array_equal(d['aperiodicity'], src['aperiodicity'])
        # against the oracle (tolerance: north_star 1e-4 relative RMS per tensor; exact vuv / frame count)
        o = oapi.encode_np(fs, x, f0_method='dio')
        assert np.array_equal(d['vuv'], o['vuv'])
        assert rel_rms(d['f0'], o['f0']) < 1e-8
        assert rel_rms(d['spectrogram'], o['spectrogram']) < 1e-8
        assert rel_rms(d['aperiodicity'], o['aperiodicity']) < 1e-8
    # decode with host-supplied noise == oracle decode with the same noise
    rng = np.random.RandomState(5)
    noise = [rng.randn(2 * len(x)) for x in xs]
    y, y_off = wb.decode_device(enc, noise=noise)
    y = y.cpu().numpy()
    for u in range(len(xs)):
        o = dict(dicts[u])
        yo = oapi.decode_np(o, noise=noise[u])['out']
        seg = y[y_off[u]:y_off[u + 1]]
        assert len(seg) == len(yo)
        assert rel_rms(seg, yo) < 1e-8
    assert wb.rt.take_flags() == [0] * 16


def test_world_encode_batch_decode_batch():
    """World.encode_batch / decode_batch (single process = one shard): same dicts as per-utterance encode(), and the
    batched decode of those dicts equals the per-utterance Requiem decode chain."""
    import random

    from world import main
    from world import synthesisRequiem as sr
    from world._synthetic import synth_utterance

    fs = 16000
    xs = [synth_utterance(50 + i, fs, 0.5 + 0.25 * i) for i in range(3)]
    W = main.World()
    dats = W.encode_batch(fs, xs, f0_method='dio', is_requiem=True)
    assert len(dats) == 3 and dats[0]['_batch_range'] == (0, 3)
    for x, d in zip(xs, dats):
        one = W.encode(fs, x, f0_method='dio', is_requiem=True)
        for key in ('f0', 'vuv', 'temporal_positions', 'spectrogram', 'aperiodicity'):
            assert np.array_equal(d[key], one[key]), key
    random.seed(1)
    np.random.seed(1)
    from world.get_seeds_signals import get_seeds_signals
    seeds = get_seeds_signals(fs)
    W.decode_batch(dats, seeds=seeds)
    sr.generate_noise.current_index = None
    for d in dats:
        y = sr.synthesisRequiem(d, d, seeds)
        m = np.max(np.abs(y))
        y = y / m if m > 1.0 else y
        assert len(d['out']) == len(y)
        assert rel_rms(d['out'], y) < 1e-10


def test_encode_batch_dicts_are_lazy_and_aliasing_holds():
    """World.encode_batch returns world.batch.EncodingDict's: dense values stay in HBM until read;
    decode_batch takes unread ones from the resident encoding and uploads what the caller read, edited or replaced.
    The resynthesis flow encode_batch -> scale_pitch -> scale_duration -> decode_batch must give the audio of the same
    flow on fully materialised plain dicts, and every reference-style way of editing a dict must take effect
    (world/main.py:154-196: in-place `*=`, `dat['spectrogram'][...] = v`, warp_spectrum's `spec[:] = warped`)."""
    from world import main
    from world._synthetic import synth_utterance
    from world.batch import EncodingDict

    fs = 16000
    xs = [synth_utterance(57 + i, fs, 0.5 + 0.2 * i) for i in range(3)]
    W = main.World()

    def flow(dats):
        for d in dats:
            W.scale_pitch(d, 1.5)
            W.scale_duration(d, 2.0)
        return W.decode_batch(dats, seed=4)

    lazy = W.encode_batch(fs, xs, f0_method='dio')
    assert all(type(d) is EncodingDict for d in lazy)
    eager = [dict(d) for d in W.encode_batch(fs, xs, f0_method='dio')]   # dict(): everything downloaded, plain dicts
    assert all(type(d) is dict and isinstance(d['spectrogram'], np.ndarray) for d in eager)
    flow(lazy)
    flow(eager)
    for a, b in zip(lazy, eager):
        assert a.resident_rows('spectrogram', W_rt()) is not None         # never read: never crossed PCIe
        assert a.resident_rows('aperiodicity', W_rt()) is not None
        assert len(a['out']) == len(b['out']) == len(np.arange(0, b['temporal_positions'][-1] + 1 / fs, 1 / fs))
        assert np.allclose(a['out'], b['out'], atol=1e-12)
        assert np.array_equal(a['f0'], b['f0']) and np.array_equal(a['temporal_positions'], b['temporal_positions'])
        a['out'][:10] = 0.0                                                # 'out' is the caller's own writeable array
    assert np.all(lazy[0]['out'][:10] == 0.0) and np.any(lazy[1]['out'][10:2000] != 0.0)
    # edits of dense values reach the decode: in place after a read, through warp_spectrum, and by replacement
    base = W.decode_batch(W.encode_batch(fs, xs, f0_method='dio'), seed=4)
    edited = W.encode_batch(fs, xs, f0_method='dio')
    plain = [dict(d) for d in W.encode_batch(fs, xs, f0_method='dio')]
    edited[0]['spectrogram'][...] = edited[0]['spectrogram'] * 4.0        # read, then edited in place
    plain[0]['spectrogram'][...] = plain[0]['spectrogram'] * 4.0
    W.warp_spectrum(edited[1], 1.1)
    W.warp_spectrum(plain[1], 1.1)
    edited[2]['aperiodicity'] = np.full_like(plain[2]['aperiodicity'], 0.5)  # replaced without ever being read
    plain[2]['aperiodicity'] = np.full_like(plain[2]['aperiodicity'], 0.5)
    W.decode_batch(edited, seed=4)
    W.decode_batch(plain, seed=4)
    for u in range(3):
        assert np.allclose(edited[u]['out'], plain[u]['out'], atol=1e-12), u
        assert not np.allclose(edited[u]['out'], base[u]['out'], atol=1e-6), u  # the edit was heard
    # utterances 0 and 1 kept their untouched aperiodicity on the device, utterance 2 its spectrogram
    assert edited[0].resident_rows('aperiodicity', W_rt()) is not None
    assert edited[2].resident_rows('spectrogram', W_rt()) is not None
    assert edited[0].resident_rows('spectrogram', W_rt()) is None


def test_facade_batches_in_two_parts_equal_the_single_batch(monkeypatch):
    """World.encode_batch / decode_batch cut a large batch into two parts on two pipelines (one part's PCIe transfer under
    the other's kernels).  Forced here on a small batch: the dicts and the audio are those of the single batch bit for
    bit — encode and the overlap-add number their work per utterance, and the Philox stream of an utterance is re-keyed
    to its index in the whole list — whether the dicts stayed resident, were materialised, or came from one encoding."""
    from world import main
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(80 + i, fs, 0.4 + 0.15 * i) for i in range(5)]
    W = main.World()

    def flow(dats, **kw):
        for d in dats:
            W.scale_pitch(d, 1.25)
            W.scale_duration(d, 1.5)
        return W.decode_batch(dats, seed=9, **kw)

    monkeypatch.setattr(main, "FACADE_SPLIT_BYTES", 1 << 60)
    assert main._decode_groups(W.encode_batch(fs, xs, f0_method='dio'), {}) == [(0, 5)]
    whole = flow(W.encode_batch(fs, xs, f0_method='dio'))
    whole_plain = [dict(d) for d in W.encode_batch(fs, xs, f0_method='dio')]
    monkeypatch.setattr(main, "FACADE_SPLIT_BYTES", 0)
    parts = W.encode_batch(fs, xs, f0_method='dio')
    assert len({id(d._enc) for d in parts}) == 2 and parts[0]['_batch_range'] == (0, 5)
    groups = main._decode_groups(parts, {})
    assert len(groups) == 2 and groups[0][0] == 0 and groups[0][1] == groups[1][0] and groups[1][1] == 5
    for a, b in zip(parts, whole_plain):
        for key in ('f0', 'vuv', 'temporal_positions'):
            assert np.array_equal(a[key], b[key]), key
    flow(parts)
    for u, (a, b) in enumerate(zip(parts, whole)):
        assert a.resident_rows('spectrogram', W_rt()) is not None      # still never read
        assert np.array_equal(a['out'], b['out']), u
        assert np.array_equal(a['spectrogram'], whole_plain[u]['spectrogram'])
        assert np.array_equal(a['aperiodicity'], whole_plain[u]['aperiodicity'])
    # plain dicts (everything uploaded again) and the lazy dicts of ONE encoding take the balanced split
    plain = flow([dict(d) for d in whole_plain])
    one = WorldBatch().encode(xs, fs, f0_method='dio').to_dicts(lazy=True)
    assert len(main._decode_groups(one, {})) == 2
    flow(one)
    rng = np.random.RandomState(3)
    noise = [rng.randn(4 * len(x)) for x in xs]
    with_noise = flow(W.encode_batch(fs, xs, f0_method='dio'), noise=noise)
    monkeypatch.setattr(main, "FACADE_SPLIT_BYTES", 1 << 60)
    with_noise_whole = flow(W.encode_batch(fs, xs, f0_method='dio'), noise=noise)
    for u in range(5):
        assert np.array_equal(plain[u]['out'], whole[u]['out']), u
        assert np.array_equal(one[u]['out'], whole[u]['out']), u
        assert np.array_equal(with_noise[u]['out'], with_noise_whole[u]['out']), u
        assert not np.array_equal(with_noise[u]['out'], whole[u]['out']), u


def test_facade_parts_retry_an_overflow_of_the_default_pulse_capacity(monkeypatch):
    """A part whose pulses overflow the default capacity (mean f0 above fs/8) is rendered again with the safe one, as
    decode_device(check=True) does for a single batch; an explicit, too small ``pulse_cap`` raises."""
    from world import _hip, main
    from world._synthetic import synth_utterance

    fs = 16000
    xs = [synth_utterance(90 + i, fs, 0.5) for i in range(4)]
    W = main.World()
    monkeypatch.setattr(main, "FACADE_SPLIT_BYTES", 1 << 60)
    ref = W.encode_batch(fs, xs, f0_method='dio')
    for d in ref:
        d['f0'][:] = 3000.0
        d['vuv'][:] = 1.0
    W.decode_batch(ref, seed=2)
    monkeypatch.setattr(main, "FACADE_SPLIT_BYTES", 0)
    hot = W.encode_batch(fs, xs, f0_method='dio')
    for d in hot[2:]:                       # only the second part overflows
        d['f0'][:] = 3000.0
        d['vuv'][:] = 1.0
    W.decode_batch(hot, seed=2)
    for u in (2, 3):
        assert np.array_equal(hot[u]['out'], ref[u]['out']), u
    with pytest.raises(_hip.WorldHipError):
        W.decode_batch(hot, seed=2, pulse_cap=64)
    for lane in (0, main.FACADE_LANE, main.FACADE_LANE + 1):
        assert _hip.Runtime.get(None, lane).take_flags() == [0] * 16   # nothing left standing for the next batch
    W.decode_batch(W.encode_batch(fs, xs, f0_method='dio'), seed=2)
    # a part that cannot even be enqueued (a spectrogram of another size in the second half): the first part's work is
    # waited for and its conditions dropped before the error reaches the caller
    bad = W.encode_batch(fs, xs, f0_method='dio')
    bad[3]['spectrogram'] = np.zeros((17, len(bad[3]['f0'])))
    with pytest.raises(Exception):
        W.decode_batch(bad, seed=2)
    for lane in (main.FACADE_LANE, main.FACADE_LANE + 1):
        assert _hip.Runtime.get(None, lane).take_flags() == [0] * 16
    good = W.decode_batch(W.encode_batch(fs, xs, f0_method='dio'), seed=2)
    assert all(np.all(np.isfinite(d['out'])) for d in good)


def W_rt():
    from world import _hip
    return _hip.Runtime.get()


def test_encode_batch_defaults_are_encodes():
    """encode_batch(fs, xs) without keywords = encode(fs, x) without keywords: Harvest (world/main.py:106)."""
    from world import main
    from world._synthetic import synth_utterance

    fs = 16000
    xs = [synth_utterance(55 + i, fs, 0.5 + 0.2 * i) for i in range(2)]
    W = main.World()
    dats = W.encode_batch(fs, xs)
    for x, d in zip(xs, dats):
        one = W.encode(fs, x)
        assert np.array_equal(d['vuv'], one['vuv'])
        assert np.array_equal(d['f0'], one['f0'])
        assert set(one) - set(d) == {'ps spectrogram'}


def test_swipe_batch_runs_on_swipes_own_grid():
    """f0_method='swipe' in a batch: frame times are swipe()'s arange * 0.005 and another frame_period is refused
    (the reference ignores frame_period on that path, world/main.py:134-135)."""
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.swipe import swipe

    fs = 16000
    x = synth_utterance(58, fs, 0.7)
    wb = WorldBatch()
    enc = wb.encode([x], fs, f0_method='swipe')
    one = swipe(fs, x, [71, 800], sTHR=0.3)
    assert np.array_equal(enc.temporal_positions.cpu().numpy(), one["temporal_positions"])
    # World.encode ignores frame_period for swipe (world/main.py:134-135): so does the batch encode
    enc4 = wb.encode([x], fs, f0_method='swipe', frame_period=4)
    assert enc4.frame_period == 5
    assert np.array_equal(enc4.temporal_positions.cpu().numpy(), one["temporal_positions"])
    assert np.array_equal(enc4.f0.cpu().numpy(), enc.f0.cpu().numpy())
    # ... but a caller-supplied batch grid of another period cannot be swipe's grid
    batch, x_d, tp_d = wb.upload([x], fs, frame_period=4)
    with pytest.raises(ValueError):
        wb.encode_device(batch, x_d, tp_d, fs, f0_method='swipe', frame_period=4)


def test_download_async_double_buffer():
    """WorldBatch.download_async: results of consecutive steps leave through alternating pinned slots on a private
    copy stream; what arrives equals a plain .cpu() of the same tensors, slot re-use waits for the previous copy."""
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(60 + i, fs, 0.5 + 0.1 * i) for i in range(3)]
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, fs)
    refs, got = [], []
    for k in range(4):
        enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio")
        y, _ = wb.decode_device(enc, seed=k)
        pins, ev = wb.download_async((enc.f0, enc.spectrogram, y), slot=k % 2)
        refs.append([t.cpu().numpy().copy() for t in (enc.f0, enc.spectrogram, y)])
        WorldBatch.download_wait(ev)
        got.append([p.numpy().copy() for p in pins])
    for r, g in zip(refs, got):
        for a, b in zip(r, g):
            assert np.array_equal(a, b)
    assert not np.array_equal(got[0][2], got[1][2])  # different noise seeds: the slots really carried different steps


def test_prefetched_time_base_equals_inline_decode():
    """encode_device computes the decode's time base on a second stream from the F0 stage's output (read through the
    rule CheapTrick / D4C apply to f0); decode_device then only renders.  Same audio as the in-line decode (up to the
    order of the overlap-add atomics); a modifier drops the prefetch; a second batch invalidates the first one's."""
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(80 + i, fs, 0.6 + 0.2 * i) for i in range(3)]
    for method in ("dio", "harvest"):
        a = WorldBatch(prefetch_timebase=True)
        b = WorldBatch(prefetch_timebase=False)
        ea = a.encode(xs, fs, f0_method=method)
        eb = b.encode(xs, fs, f0_method=method)
        assert ea._timebase is not None and eb._timebase is None
        assert ea.timebase_for(a, None) is not None
        for t in ("f0", "vuv", "spectrogram", "aperiodicity"):
            assert np.array_equal(getattr(ea, t).cpu().numpy(), getattr(eb, t).cpu().numpy())
        ya, offa = a.decode_device(ea, seed=4)
        yb, offb = b.decode_device(eb, seed=4)
        assert np.array_equal(offa, offb)
        ya, yb = ya.cpu().numpy(), yb.cpu().numpy()
        assert np.max(np.abs(ya - yb)) <= 1e-15 * max(1.0, np.max(np.abs(yb)))
        # decoding again re-uses the same time base; host noise goes through it too
        rng = np.random.RandomState(1)
        noise = [rng.randn(2 * len(x)) for x in xs]
        y2 = a.decode_device(ea, noise=noise)[0].cpu().numpy()
        y3 = b.decode_device(eb, noise=noise)[0].cpu().numpy()
        assert np.max(np.abs(y2 - y3)) <= 1e-15 * max(1.0, np.max(np.abs(y3)))
        # a modifier invalidates it (the time base depends on f0 and the frame times)
        ea.scale_pitch(1.5)
        eb.scale_pitch(1.5)
        assert ea.timebase_for(a, None) is None
        y4 = a.decode_device(ea, seed=4)[0].cpu().numpy()
        y5 = b.decode_device(eb, seed=4)[0].cpu().numpy()
        assert np.max(np.abs(y4 - y5)) <= 1e-15 * max(1.0, np.max(np.abs(y5)))
        # another encode on the same WorldBatch takes the time-base context over
        e1 = a.encode(xs, fs, f0_method=method)
        e2 = a.encode(xs[:2], fs, f0_method=method)
        assert e1.timebase_for(a, None) is None and e2.timebase_for(a, None) is not None
        y6 = a.decode_device(e1, seed=4)[0].cpu().numpy()  # falls back to the in-line time base
        assert np.max(np.abs(y6 - yb)) <= 1e-15 * max(1.0, np.max(np.abs(yb)))
        assert a.rt.take_flags() == [0] * 16


def test_encode_batch_ps_spectrogram_opt_in(golden):
    """encode() returns CheapTrick's complex spectra as 'ps spectrogram' (world/main.py:149, world/cheaptrick.py:30,38);
    the batch path keeps them on request: against the reference fixture and against the single-utterance encode()."""
    from world import main
    from world._synthetic import synth_utterance

    g = golden("getters")
    fs = int(g["fs"])
    x = synth_utterance(int(g["utt"]), fs, float(g["seconds"]))  # the fixture's input (tests/test_hip_getters.py)
    other = synth_utterance(61, fs, 0.4)
    w = main.World()
    dats = w.encode_batch(fs, [x, other], f0_method='dio', want_ps=True)
    assert dats[0]['ps spectrogram'].shape == tuple(g["getspec_ps_shape"])
    assert dats[0]['ps spectrogram'].dtype == np.complex128
    ps = dats[0]['ps spectrogram'][:, g["getspec_ps_cols"]]
    ref = g["getspec_ps"]
    assert np.sqrt(np.mean(np.abs(ps - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 1e-10
    one = w.encode(fs, other, f0_method='dio')
    assert np.array_equal(dats[1]['ps spectrogram'], one['ps spectrogram'])
    assert 'ps spectrogram' not in w.encode_batch(fs, [other], f0_method='dio')[0]
    from world.batch import WorldBatch
    with pytest.raises(ValueError):
        WorldBatch().encode([other], fs, f0_method='dio').to_dicts(want_ps=True)


def test_deferred_flag_check_reports_late_but_not_never():
    """check='deferred': no host wait in encode/decode; a condition raised by a kernel is published behind the call's
    work (wh_flags_post) and raised by the next deferred-check call, or by check().  Driven with a pulse capacity that
    is too small (WH_FLAG_PULSE_OVERFLOW)."""
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(62, fs, 0.6), synth_utterance(63, fs, 0.5)]
    wb = WorldBatch(prefetch_timebase=False)
    enc = wb.encode(xs, fs, f0_method='dio', check='deferred')
    y_ok, off = wb.decode_device(enc, check='deferred')      # nothing to report
    wb.decode_device(enc, pulse_cap=8, check='deferred')     # overflows: returns without raising ...
    wb.rt.torch.cuda.synchronize()
    with pytest.raises(_hip.WorldHipError, match="pulse_cap"):
        wb.decode_device(enc, check='deferred')              # ... the next deferred call does
    y2, _ = wb.decode_device(enc, check='deferred')          # reported once, then clean again
    wb.check()
    assert np.array_equal(y2.cpu().numpy(), y_ok.cpu().numpy()) or np.allclose(y2.cpu().numpy(), y_ok.cpu().numpy(), atol=1e-13)
    # check() (synchronising) also sees what was posted and not yet polled
    wb.decode_device(enc, pulse_cap=8, check='deferred')
    with pytest.raises(_hip.WorldHipError, match="pulse_cap"):
        wb.check()
    assert wb.rt.take_flags() == [0] * 16


@pytest.mark.parametrize("mode", ["deferred", False])
def test_prefetched_timebase_conditions_survive_the_next_prefetch(mode, monkeypatch):
    """encode A (time base prefetched), decode A (asynchronous check), encode B.  B's prefetch used to clear
    the shared time-base context's flags on the assumption that they belonged to a superseded prefetch — but A's time
    base had been rendered, so its PULSE_OVERFLOW was still owed to the caller: truncated audio, no error.  The
    default capacity is forced down to 8 pulses so that the prefetched time base itself overflows."""
    from world import _hip, batch as wbatch
    from world._synthetic import synth_utterance

    fs = 16000
    xs = [synth_utterance(66, fs, 0.5), synth_utterance(67, fs, 0.4)]
    wb = wbatch.WorldBatch(prefetch_timebase=True)
    wb.check()  # (the time-base context is shared per device and lane: start clean)
    monkeypatch.setattr(wbatch, "default_pulse_cap", lambda ny: 8)
    enc_a = wb.encode(xs, fs, f0_method='dio', check=mode)
    assert enc_a._timebase is not None and enc_a._timebase["pulse_cap"] == 8
    wb.decode_device(enc_a, check=mode)           # renders from the overflowing time base; returns without raising
    monkeypatch.undo()
    if mode == 'deferred':
        with pytest.raises(_hip.WorldHipError, match="pulse_cap"):
            for _ in range(3):                    # late (the post may not have executed at the first poll), never lost
                wb.encode(xs, fs, f0_method='dio', check=mode)
                wb.rt.torch.cuda.synchronize()
    else:
        wb.encode(xs, fs, f0_method='dio', check=mode)   # the next prefetch must not clear A's condition
        with pytest.raises(_hip.WorldHipError, match="pulse_cap"):
            wb.check()
    wb.rt.torch.cuda.synchronize()
    try:
        wb.check()
    except _hip.WorldHipError:
        pass
    # a time base that nobody rendered from takes its conditions with it: nothing is blamed on the next batch
    monkeypatch.setattr(wbatch, "default_pulse_cap", lambda ny: 8)
    wb.encode(xs, fs, f0_method='dio', check=False)
    monkeypatch.undo()
    enc_c = wb.encode(xs, fs, f0_method='dio', check=False)
    wb.decode_device(enc_c, check=False)
    wb.check()


def test_timebase_lasts_until_the_next_workspace_call_of_its_context():
    """C-ABI contract (include/world_hip.h, INTEGRATION.md "One context, one time base"): the time base of
    wh_synthesis_timebase lives in its context's workspace and ANY later workspace-using call on that context drops it —
    wh_synthesis_render then fails loudly instead of rendering from a workspace that has been laid out again."""
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.synthesis import default_pulse_cap, synthesis_device, synthesis_timebase_device, time_axis_params

    fs = 16000
    xs = [synth_utterance(68, fs, 0.4)]
    wb = WorldBatch(prefetch_timebase=False)
    enc = wb.encode(xs, fs, f0_method='dio')
    rt = wb.rt
    geo = [time_axis_params(enc.host_times(), fs)]
    ny, t0, dt = [geo[0][0]], [geo[0][1]], [geo[0][2]]
    cap = default_pulse_cap(ny)
    args = (enc.batch, enc.temporal_positions, enc.f0, enc.vuv, enc.spectrogram, enc.aperiodicity, fs, enc.fft_size, ny, t0, dt)
    with rt.on_stream():
        synthesis_timebase_device(rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, fs, ny, t0, dt, cap)
        y1, _ = synthesis_device(rt, *args, seed=5, pulse_cap=cap, timebase_rt=rt)   # right behind it: fine
        y1b, _ = synthesis_device(rt, *args, seed=5, pulse_cap=cap, timebase_rt=rt)  # (a render reserves nothing: again)
        with pytest.raises(_hip.WorldHipError, match="no matching time base"):       # another pulse capacity: not this one
            synthesis_device(rt, *args, seed=5, pulse_cap=cap + 1, timebase_rt=rt)
        wb._peak_normalise(y1.clone(), np.array([0, ny[0]], dtype=np.int64))                 # any workspace-using stage call
        with pytest.raises(_hip.WorldHipError, match="no matching time base"):
            synthesis_device(rt, *args, seed=5, pulse_cap=cap, timebase_rt=rt)
        y2, _ = synthesis_device(rt, *args, seed=5, pulse_cap=cap)                   # the whole synthesis: as before
    assert np.allclose(y1.cpu().numpy(), y2.cpu().numpy(), atol=1e-12)
    assert np.allclose(y1b.cpu().numpy(), y2.cpu().numpy(), atol=1e-12)
    wb.check()


def test_ctx_trim_gives_scratch_back_and_the_next_call_allocates_again():
    """wh_ctx_trim (ABI 104): the arena and the per-call buffers of a context only grow; trimming frees them (device memory
    in use drops), drops a held time base, and the next calls — same batch, so the cached per-call tables would have been
    taken as resident — allocate, upload and compute the same results again."""
    import torch
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(70 + i, fs, 1.0 + 0.3 * i) for i in range(3)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method='harvest', is_requiem=True)
    y, _ = wb.decode_device(enc, cursor=np.zeros(3))
    enc2 = wb.encode(xs, fs, f0_method='dio')
    y2, _ = wb.decode_device(enc2, seed=2)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    assert enc2._timebase is not None
    _hip.Runtime.trim_all()
    assert torch.cuda.mem_get_info()[0] > free_before  # (the Harvest arena alone is tens of MB at this size)
    assert enc2.timebase_for(wb, None) is None          # the prefetched time base went with its context's arena:
    y2b, _ = wb.decode_device(enc2, seed=2)             # the decode computes it in line
    assert np.array_equal(y2b.cpu().numpy(), y2.cpu().numpy())
    again = wb.encode(xs, fs, f0_method='harvest', is_requiem=True)
    for name in ("f0", "vuv", "spectrogram", "aperiodicity"):
        assert torch.equal(getattr(again, name), getattr(enc, name)), name
    ya, _ = wb.decode_device(again, cursor=np.zeros(3))
    assert torch.equal(ya, y)
    wb.check()


def test_encode_device_under_inference_mode():
    """Tensors made under torch.inference_mode() have no version counter: the time-base prefetch is skipped, the encode
    and decode work as without it."""
    import torch
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(64, fs, 0.5)]
    wb = WorldBatch()
    ref = wb.encode(xs, fs, f0_method='dio')
    y_ref, _ = wb.decode_device(ref, seed=3)
    with torch.inference_mode():
        enc = wb.encode(xs, fs, f0_method='dio')
        y, _ = wb.decode_device(enc, seed=3)
    assert np.array_equal(enc.f0.cpu().numpy(), ref.f0.cpu().numpy())
    assert np.allclose(y.cpu().numpy(), y_ref.cpu().numpy(), atol=1e-13)
