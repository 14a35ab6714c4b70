"""GPU: BASELINE config 2 at FULL size (64 x 10 s, 16 kHz) checked through size-independent properties
(the oracle would need minutes here): exact frame / sample counts, batch == single-utterance results,
utterance-permutation equivariance, input-gain scaling laws, and waveform-level resynthesis sanity."""
import numpy as np
import pytest

from conftest import synth_cached

pytestmark = pytest.mark.gpu

FS = 16000


@pytest.fixture(scope="module")
def full_batch():
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    xs = [synth_cached(u, FS, 10.0) for u in range(64)]
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="dio")
    return xs, wb, enc


def test_counts_and_flags(full_batch):
    xs, wb, enc = full_batch
    assert enc.batch.total_frames == 64 * 2001          # F = int(1000*N/fs/5 + 1), bit-exact
    y, y_off = wb.decode_device(enc, seed=7)
    assert list(np.diff(y_off)) == [160001] * 64        # len(np.arange(0, tp[-1]+1/fs, 1/fs))
    assert wb.rt.take_flags() == [0] * 16
    y = y.cpu().numpy()
    assert np.all(np.isfinite(y)) and np.max(np.abs(y)) <= 1.0 + 1e-12   # peak normalisation (main.py:209-212)
    f0 = enc.f0.cpu().numpy()
    vuv = enc.vuv.cpu().numpy()
    assert np.all((f0 == 0) == (vuv == 0)) or np.all(f0[vuv == 0] == 0)   # d4c zeroes unvoiced f0 (Q6)
    voiced = f0[(vuv != 0) & (f0 != 500.0)]
    assert 50 < voiced.min() and voiced.max() < 1000   # DIO candidates in [71, 800], StoneMask may move them by <= 20 %
    sp = enc.spectrogram
    ap = enc.aperiodicity
    assert bool((sp > 0).all()) and bool(((ap > 0) & (ap <= 1)).all())   # a 0 dB band gives exactly 1.0, like the reference


def test_batch_rows_equal_single_utterance(full_batch):
    """No cross-utterance state: utterance u inside the 64-batch == the same utterance alone (bitwise)."""
    from world.batch import WorldBatch

    xs, wb, enc = full_batch
    fo = enc.batch.frame_off
    for u in (0, 17, 63):
        single = WorldBatch().encode([xs[u]], FS, f0_method="dio")
        s = slice(int(fo[u]), int(fo[u + 1]))
        for name in ("f0", "vuv", "spectrogram", "aperiodicity"):
            a = getattr(enc, name)[s].cpu().numpy()
            b = getattr(single, name).cpu().numpy()
            assert np.array_equal(a, b), (u, name)


def test_permutation_equivariance(full_batch):
    from world.batch import WorldBatch

    xs, wb, enc = full_batch
    perm = np.random.RandomState(0).permutation(8)
    sub = [xs[i] for i in range(8)]
    e1 = WorldBatch().encode(sub, FS, f0_method="dio")
    e2 = WorldBatch().encode([sub[i] for i in perm], FS, f0_method="dio")
    f1 = e1.spectrogram.cpu().numpy().reshape(8, 2001, -1)
    f2 = e2.spectrogram.cpu().numpy().reshape(8, 2001, -1)
    assert np.array_equal(f2, f1[perm])


def test_gain_scaling_laws(full_batch):
    """x -> 0.5*x: F0, VUV and aperiodicity are scale-free, the power-spectral envelope scales by 0.25
    (exact up to rounding because 0.5 is a power of two)."""
    from world.batch import WorldBatch

    xs, wb, enc = full_batch
    sub = xs[:4]
    e1 = WorldBatch().encode(sub, FS, f0_method="dio")
    e2 = WorldBatch().encode([0.5 * x for x in sub], FS, f0_method="dio")
    assert np.array_equal(e1.vuv.cpu().numpy(), e2.vuv.cpu().numpy())
    assert np.allclose(e1.f0.cpu().numpy(), e2.f0.cpu().numpy(), rtol=1e-12, atol=0)
    assert np.allclose(e1.aperiodicity.cpu().numpy(), e2.aperiodicity.cpu().numpy(), rtol=0, atol=1e-9)
    s1, s2 = e1.spectrogram.cpu().numpy(), e2.spectrogram.cpu().numpy()
    # exact up to the eps/2 log-guard (not scale-free): visible only on ~1e-9-level bins
    assert np.allclose(s2, 0.25 * s1, rtol=1e-6, atol=0)
    assert np.sqrt(np.mean((s2 - 0.25 * s1) ** 2) / np.mean(s2 ** 2)) < 1e-12


def test_resynthesis_tracks_input(full_batch):
    """encode → decode reproduces the utterance's short-time energy envelope (voiced/unvoiced alternation)."""
    xs, wb, enc = full_batch
    y, y_off = wb.decode_device(enc, seed=3)
    y = y.cpu().numpy()
    for u in (0, 31):
        seg = y[y_off[u]:y_off[u + 1]][:160000]
        ex = np.sqrt(np.mean(xs[u].reshape(-1, 1600) ** 2, axis=1))
        ey = np.sqrt(np.mean(seg.reshape(-1, 1600) ** 2, axis=1))
        assert np.corrcoef(ex, ey)[0, 1] > 0.9


def test_decoded_rows_equal_the_single_utterance_decode_bitwise(full_batch):
    """The pulse-wise decode of config 2 at its full size, sample by sample: row u of the 64 x 10 s batch == the same
    utterance decoded alone under ``philox_seed_for_offset(seed, u)`` — the seed under which utterance 0 of a batch draws
    the noise utterance u of the whole batch draws.  The overlap-add (response_kernel's runs of pulses +
    response_gather_kernel) numbers its work per utterance and adds in a fixed order, so this is an equality of bits;
    the reference adds pulse after pulse into one array (world/synthesis.py:61-81)."""
    from world.batch import WorldBatch
    from world.synthesis import philox_seed_for_offset

    xs, wb, enc = full_batch
    y, y_off = wb.decode_device(enc, seed=7)
    fo = enc.batch.frame_off
    for u in (0, 1, 37, 63):
        single = WorldBatch().encode([xs[u]], FS, f0_method="dio")
        ys, _ = WorldBatch().decode_device(single, seed=philox_seed_for_offset(7, u))
        a = y[int(y_off[u]):int(y_off[u + 1])]
        assert a.shape == ys.shape and bool((a == ys).all()), u
        assert float(a.abs().max()) > 0.05
    # (a row decoded alone under the batch's own seed is utterance 0's stream: another noise)
    single = WorldBatch().encode([xs[37]], FS, f0_method="dio")
    other, _ = WorldBatch().decode_device(single, seed=7)
    assert not bool((other == y[int(y_off[37]):int(y_off[38])]).all())


def test_harvest_batch_rows_equal_single_utterance():
    """Harvest on a ragged batch of 24 utterances of 9-10 s (config 3's shape at a size every stage's grid is busy:
    utterance-fastest band walkers, 16-frame refinement / pruning blocks that straddle utterance ends): every utterance
    inside the batch == the same utterance alone, bitwise; no device flag."""
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    xs = [synth_utterance(100 + u, FS, 9.0 + 0.043 * u) for u in range(24)]
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="harvest")
    assert wb.rt.take_flags() == [0] * 16
    fo = enc.batch.frame_off
    for u in (0, 11, 23):
        single = WorldBatch().encode([xs[u]], FS, f0_method="harvest")
        s = slice(int(fo[u]), int(fo[u + 1]))
        for name in ("f0", "vuv", "spectrogram", "aperiodicity"):
            a = getattr(enc, name)[s].cpu().numpy()
            b = getattr(single, name).cpu().numpy()
            assert np.array_equal(a, b), (u, name)
    f0 = enc.f0.cpu().numpy()
    vuv = enc.vuv.cpu().numpy()
    voiced = f0[vuv != 0]
    assert len(voiced) > 0.5 * len(f0) and 60 < voiced.min() and voiced.max() < 900


def test_facade_at_96k(golden):
    """World().encode(harvest) + seeded decode at 96 kHz against the reference's stage outputs (golden_syn96k.npz):
    Harvest exact in vuv, CheapTrick 4096-point, D4C 8192-point, pulse-wise synthesis 4096-point."""
    import numpy as np
    from conftest import rel_rms
    from world import main

    g = golden("syn96k")
    fs = int(g["fs"])
    W = main.World()
    dat = W.encode(fs, g["x"].copy(), f0_method="harvest")
    assert np.array_equal(dat["vuv"], g["harvest_vuv"])
    assert np.max(np.abs(dat["f0"] - g["d4c_f0_after"])) < 1e-6
    assert dat["spectrogram"].shape == g["ct_spectrogram"].shape
    assert rel_rms(dat["spectrogram"], g["ct_spectrogram"]) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"] - g["d4c_aperiodicity"])) < 1e-6
    np.random.seed(int(g["seed"]))
    y = W.decode(dict(dat))["out"]
    ref = g["syn_y"] / max(1.0, np.max(np.abs(g["syn_y"])))
    assert len(y) == len(ref) and rel_rms(y, ref) < 1e-6
    # Requiem path at this rate too (band aperiodicity from an 8192-point transform; 2048-point seed pulses)
    datr = W.encode(fs, g["x"].copy(), f0_method="harvest", is_requiem=True)
    assert np.max(np.abs(datr["aperiodicity"] - g["req_band_ap"])) < 1e-5
    yr = W.decode(dict(datr))["out"]
    assert np.all(np.isfinite(yr)) and len(yr) == len(ref)
