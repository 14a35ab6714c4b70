"""GPU parity of the spectral feature heads (world/main.py:275-365) against reference output
(tests/golden/golden_heads.npz, make_golden.py heads_fixture): log mel filterbank energies and both cepstral
transforms through the FP64-MFMA product kernel, context stacking, and the same heads on a resident batch."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _spec(golden):
    return np.ascontiguousarray(golden("syn16k")["ct_spectrogram"].T)


def test_encode_lfbank(golden):
    from world import main

    g = golden("heads")
    W = main.World()
    spec = _spec(golden)
    lf = W.encode_lfbank(spec)
    assert lf.shape == g["lfbank"].shape
    assert np.max(np.abs(lf - g["lfbank"])) < 1e-11  # log energies: absolute
    lf2 = W.encode_lfbank(spec, prefac=0.9, nfilt=24, lowfreq=100, highfreq=6000)
    assert np.max(np.abs(lf2 - g["lfbank_24_hi6k"])) < 1e-11
    # an all-zero frame hits the 0 -> eps substitution (main.py:321)
    z = W.encode_lfbank(np.zeros((3, 513)))
    assert np.all(z == np.log(np.finfo(float).eps))


def test_encode_decode_mcep(golden):
    from world import main

    g = golden("heads")
    W = main.World()
    spec = _spec(golden)
    mc = W.encode_mcep(spec)
    assert mc.shape == g["mcep"].shape
    assert np.max(np.abs(mc - g["mcep"])) < 1e-12
    assert np.max(np.abs(W.encode_mcep(spec, n0=20) - g["mcep_20"])) < 1e-12
    dec = W.decode_mcep(g["mcep"], 1024)
    assert dec.shape == (spec.shape[0], 513)
    assert rel_rms(dec[::40], g["imcep_rows"]) < 1e-12
    assert rel_rms(dec.sum(axis=0), g["imcep_colsum"]) < 1e-12
    assert rel_rms(dec.sum(axis=1), g["imcep_rowsum"]) < 1e-12


def test_get_context_and_tables(golden):
    from world import main

    g = golden("heads")
    W = main.World()
    assert np.array_equal(W.get_filterbanks(), g["fbank_20_512"])
    assert np.array_equal(W.get_filterbanks(32, 1024, 16000).sum(axis=1), g["fbank_32_1024_rowsum"])
    ctx = W.get_context(g["lfbank"], w=5)
    assert ctx.shape == tuple(g["context_shape"])
    assert np.array_equal(ctx[[0, 1, 4, 5, 120, 235, 236, 240]], g["context_rows"])
    assert np.array_equal(W.get_context(g["mcep"], w=2).sum(axis=1), g["context_w2_rowsum"])


def test_heads_on_resident_batch():
    """BatchEncoding.lfbank / mcep read the frame-major device spectrogram in place: ragged batch == per utterance."""
    from world import main
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(90 + i, fs, 0.4 + 0.3 * i) for i in range(3)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio")
    lf = enc.lfbank().cpu().numpy()
    mc = enc.mcep().cpu().numpy()
    W = main.World()
    fo = enc.batch.frame_off
    for u, d in enumerate(enc.to_dicts()):
        sl = slice(int(fo[u]), int(fo[u + 1]))
        assert np.array_equal(lf[sl], W.encode_lfbank(d["spectrogram"].T, fs=fs))
        assert np.array_equal(mc[sl], W.encode_mcep(d["spectrogram"].T, fs=fs))
