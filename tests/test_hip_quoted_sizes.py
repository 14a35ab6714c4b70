"""GPU: results READ at the sizes BASELINE.json quotes its targets on — until now those sizes were
only timed, with a flag read.

  * north_star / config 4: 1024 x 10 s at 16 kHz on ONE GPU, Harvest + CheapTrick + D4C-Requiem encode and Requiem
    decode, one launch per kernel over all 2 049 024 frames (the [channel][frame] raw-candidate map of that launch has
    1.56 G elements);
  * config 3: Harvest on 256 x 10 s;
  * config 5: 16 x 60 s at 48 kHz, Harvest encode, scale_pitch(1.5), scale_duration(2.0), pulse-wise decode.

The oracle would need hours at these sizes; the checks are the size-independent ones the domain offers: a row of the big
batch == the same utterance encoded alone (bitwise: no cross-utterance state, no index that overflows at scale), copies
of one utterance at different positions of the batch agree with each other, decoded rows equal the single-utterance
decode started at the chained noise cursor (world/synthesisRequiem.py:131-141), exact frame / sample counts (Q9), no
device flag, and — config 5 — the 60 s Harvest contour of a batch row against the reference's own output
(golden_longform48k.npz).  The batches tile a few distinct utterances so that host-side generation stays cheap."""
import numpy as np
import pytest

from conftest import synth_cached

pytestmark = pytest.mark.gpu

FS = 16000
NY = 160001  # len(np.arange(0, tp[-1] + 1/fs, 1/fs)) of a 10 s utterance (Q9)
NF = 2001


def _rows_equal(enc, single, u, names):
    fo = enc.batch.frame_off
    s = slice(int(fo[u]), int(fo[u + 1]))
    for name in names:
        a = getattr(enc, name)[s]
        b = getattr(single, name)
        assert a.shape == b.shape and bool((a == b).all()), (u, name)


def _copies_agree(t, copies, per_copy_rows):
    """t: per-frame tensor of a batch that repeats the same utterances `copies` times: every copy == the first."""
    v = t.reshape(copies, per_copy_rows, -1)
    for c in range(1, copies):
        assert bool((v[c] == v[0]).all()), c


def test_north_star_batch_1024x10s_rows_equal_single_utterance():
    from world.batch import WorldBatch
    from world.synthesisRequiem import _advance

    distinct, copies = 64, 16
    base = [synth_cached(u, FS, 10.0) for u in range(distinct)]
    xs = base * copies  # utterance r of the batch = base[r % 64]
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="harvest", is_requiem=True)
    assert wb.rt.take_flags() == [0] * 16
    assert enc.batch.total_frames == 1024 * NF
    assert enc.aperiodicity.shape == (1024 * NF, 3) and enc.spectrogram.shape == (1024 * NF, 513)
    y, y_off = wb.decode_device(enc)
    assert wb.rt.take_flags() == [0] * 16
    assert list(np.diff(y_off)) == [NY] * 1024
    assert bool(y.isfinite().all())
    # copies of an utterance agree wherever they sit in the batch (analysis: bitwise)
    for name in ("f0", "vuv", "spectrogram", "aperiodicity"):
        _copies_agree(getattr(enc, name), copies, distinct * NF)
    voiced = enc.vuv.reshape(1024, NF).sum(dim=1)
    assert float(voiced.min()) > 0.5 * NF  # every utterance was analysed (0.8 s voiced / 0.2 s unvoiced gating)
    # rows == the same utterance encoded and decoded alone
    nlen = None
    for r in (0, 511, 1023):
        alone = WorldBatch()
        single = alone.encode([xs[r]], FS, f0_method="harvest", is_requiem=True)
        _rows_equal(enc, single, r, ("f0", "vuv", "spectrogram", "aperiodicity", "temporal_positions"))
        if nlen is None:
            from world.synthesisRequiem import _default_seeds
            nlen = int(_default_seeds[(FS, wb.rt.index, wb.rt.lane)]["noise_d"].shape[0])
        cur = np.zeros(3)
        for _ in range(r):  # the batch hands the circular noise-seed cursor from utterance to utterance
            cur = _advance(cur, NY, nlen)
        ys, _ = alone.decode_device(single, cursor=cur)
        a = y[int(y_off[r]):int(y_off[r + 1])]
        # (peak normalisation is per utterance; the overlap-add is summed in a fixed order: bitwise — see
        # tests/test_hip_determinism.py)
        assert float((a - ys).abs().max()) <= 1e-12, r
    del y, enc
    wb.rt.torch.cuda.empty_cache()


def test_config3_harvest_256x10s_rows_equal_single_utterance():
    from world.batch import WorldBatch
    from world.harvest import harvest_device

    distinct, copies = 64, 4
    base = [synth_cached(u, FS, 10.0) for u in range(distinct)]
    xs = base * copies
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, FS)
    with wb.rt.on_stream():
        f0, vuv = harvest_device(wb.rt, batch, x_d, tp_d, FS, 71, 800, 5)
    assert wb.rt.take_flags() == [0] * 16
    assert f0.shape == (256 * NF,)
    _copies_agree(f0, copies, distinct * NF)
    _copies_agree(vuv, copies, distinct * NF)
    for r in (0, 129, 255):
        alone = WorldBatch()
        b1, x1, t1 = alone.upload([xs[r]], FS)
        with alone.rt.on_stream():
            f1, v1 = harvest_device(alone.rt, b1, x1, t1, FS, 71, 800, 5)
        s = slice(r * NF, (r + 1) * NF)
        assert bool((f0[s] == f1).all()) and bool((vuv[s] == v1).all()), r
    f0h, vuvh = f0.cpu().numpy(), vuv.cpu().numpy()
    assert np.all((f0h > 0) == (vuvh > 0))
    voiced = f0h[vuvh > 0]
    assert 60 < voiced.min() and voiced.max() < 900 and 0.6 < (vuvh > 0).mean() < 0.9


def test_config5_16x60s_48k_modified_decode(golden):
    from world.batch import WorldBatch

    fs, seconds = 48000, 60.0
    g = golden("longform48k")
    assert int(g["utt"]) == 75 and float(g["seconds"]) == seconds
    base = [synth_cached(75, fs, seconds), synth_cached(77, fs, seconds)]
    xs = base * 8  # 16 x 60 s: config 5's per-GPU share (128 utterances over 8 GPUs)
    nf = 12001
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="harvest")
    assert wb.rt.take_flags() == [0] * 16
    assert enc.batch.total_frames == 16 * nf and enc.fft_size == 2048
    assert enc.spectrogram.shape == (16 * nf, 1025) and enc.aperiodicity.shape == (16 * nf, 1025)
    for name in ("f0", "vuv", "spectrogram", "aperiodicity"):
        _copies_agree(getattr(enc, name), 8, 2 * nf)
    # a row in the middle of the batch against the REFERENCE's Harvest output for this utterance (make_golden.py
    # longform_fixture): the whole 60 s contour, 60 001 1 ms frames picked down to the 5 ms grid
    r = 10  # utterance 75
    s = slice(r * nf, (r + 1) * nf)
    assert np.array_equal(enc.temporal_positions[s].cpu().numpy(), g["tp"])
    assert np.array_equal(enc.vuv[s].cpu().numpy(), g["harvest_vuv"])
    assert np.max(np.abs(enc.f0[s].cpu().numpy() - g["harvest_f0"])) < 1e-6
    # and == the same utterance alone, bitwise
    single = WorldBatch().encode([xs[r]], fs, f0_method="harvest")
    _rows_equal(enc, single, r, ("f0", "vuv", "spectrogram", "aperiodicity"))
    # config 5's modifiers and decode: 2 x 60 s of audio per utterance (Q9: the float-arange length)
    enc.scale_pitch(1.5).scale_duration(2.0)
    single.scale_pitch(1.5).scale_duration(2.0)
    y, y_off = wb.decode_device(enc, seed=11)
    assert wb.rt.take_flags() == [0] * 16
    tp_end = float(g["tp"][-1]) * 2.0
    ny = len(np.arange(0, tp_end + 1 / fs, 1 / fs))
    assert list(np.diff(y_off)) == [ny] * 16
    assert bool(y.isfinite().all()) and float(y.abs().max()) <= 1.0 + 1e-12
    # the device noise stream is keyed by (seed, utterance index): row r decoded ALONE under the seed that makes its
    # utterance 0 draw what utterance r of the batch draws (synthesis.philox_seed_for_offset) gives the batch's row bit
    # for bit — the overlap-add sums runs of pulses numbered per utterance, in a fixed order (world/synthesis.py:61-81
    # adds them serially); every sample of the 2 x 60 s row is compared
    from world.synthesis import philox_seed_for_offset
    ys, _ = WorldBatch().decode_device(single, seed=philox_seed_for_offset(11, r))
    a = y[int(y_off[r]):int(y_off[r + 1])]
    assert a.shape == ys.shape and bool((a == ys).all())
    assert float(a.abs().max()) > 0.05  # (not a silent row)
    # and another row of the same utterance drew ANOTHER stream: equal pulse train, different noise
    b = y[int(y_off[r - 2]):int(y_off[r - 1])]
    assert not bool((a == b).all())
