"""GPU: sub-batches in flight on separate HIP streams / library contexts (WorldBatchLanes) give exactly the
results of the single-stream batch — lanes only change scheduling, never numbers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_equal_single_stream(lanes):
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch, WorldBatchLanes

    fs = 16000
    xs = [synth_utterance(40 + i, fs, 0.6 + 0.15 * (i % 3)) for i in range(5)]  # ragged, 5 utterances over 2-3 lanes
    rng = np.random.RandomState(9)
    noise = [rng.randn(2 * len(x)) for x in xs]

    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method='dio')
    ref = enc.to_dicts()
    y_ref, y_off_ref = wb.decode_device(enc, noise=noise)
    y_ref = y_ref.cpu().numpy()

    wl = WorldBatchLanes(lanes=lanes)
    wl.upload(xs, fs)
    parts = wl.split([len(x) for x in xs], lanes)
    assert parts[0][0] == 0 and parts[-1][1] == len(xs)
    assert wl.total_frames == enc.batch.total_frames
    for rep in range(2):  # second pass: contexts warm, all launches truly asynchronous
        encs = wl.encode_device(fs, f0_method='dio')
        outs = [wbl.decode_device(e, noise=noise[a:b]) for wbl, e, (a, b) in zip(wl.lanes, encs, parts)]
        wl.synchronize()
        for e, (y, y_off), (a, b) in zip(encs, outs, parts):
            dicts = e.to_dicts()
            y = y.cpu().numpy()
            for k, u in enumerate(range(a, b)):
                for key in ('f0', 'vuv', 'temporal_positions', 'spectrogram', 'aperiodicity'):
                    assert np.array_equal(dicts[k][key], ref[u][key]), (key, u, rep)
                seg = y[y_off[k]:y_off[k + 1]]
                want = y_ref[y_off_ref[u]:y_off_ref[u + 1]]
                assert len(seg) == len(want)
                # overlap-add uses FP64 atomics: the summation order of overlapping pulses is not fixed
                assert np.max(np.abs(seg - want)) <= 1e-13 * max(1.0, np.max(np.abs(want))), (u, rep)
    for wbl in wl.lanes:
        assert wbl.rt.take_flags() == [0] * 16


def test_two_batches_in_flight_equal_one_at_a_time():
    """WorldBatchPipeline: batches dealt to two independent pipelines (what bench.py --in-flight 2 times) give the
    results of one WorldBatch processing them one after the other, bit for bit — encode and (atomics-free) decode."""
    import torch
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch, WorldBatchPipeline

    fs = 16000
    batches = [[synth_utterance(400 + 10 * b + i, fs, 0.7 + 0.2 * i) for i in range(3)] for b in range(4)]
    rng = np.random.RandomState(9)
    noises = [[rng.randn(2 * len(x)) for x in xs] for xs in batches]
    one = WorldBatch()
    ref = []
    for xs, nz in zip(batches, noises):
        enc = one.encode(xs, fs, f0_method="dio")
        y, _ = one.decode_device(enc, noise=nz)
        ref.append((enc.f0.clone(), enc.spectrogram.clone(), enc.aperiodicity.clone(), y.clone()))
    pipe = WorldBatchPipeline(depth=2)
    got = [pipe.encode_decode(xs, fs, decode_kw={"noise": nz}, f0_method="dio") for xs, nz in zip(batches, noises)]
    pipe.synchronize()
    assert pipe.pipes[0].rt is not pipe.pipes[1].rt and pipe.pipes[0].rt.own_stream is not pipe.pipes[1].rt.own_stream
    for (enc, y, _), (f0, sp, ap, yr) in zip(got, ref):
        assert torch.equal(enc.f0, f0) and torch.equal(enc.spectrogram, sp) and torch.equal(enc.aperiodicity, ap)
        assert torch.equal(y, yr)

