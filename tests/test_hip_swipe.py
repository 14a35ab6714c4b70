"""GPU parity of SWIPE' (f0_method='swipe') against reference output (tests/golden/golden_swipe.npz, make_golden.py
swipe_fixture).  Voicing is exact.  A voiced frame's f0 is 2**(log2(pc) + k/768) with k the argmax of a parabola on a
1/768-octave grid; the device sums the spectral products in a different order than SciPy / BLAS, so the strengths
agree to ~1e-13 and k is identical except where two grid points tie to that precision — the test allows at most TWO
one-step differences (0.09 % in f0) per utterance and requires everything else to match to rounding.  The tone cases sit
on the candidates whose kernels carry the reference's sieve quirk (prime squares kept, world/swipe.py:158-172)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _compare(f0, vuv, ref_f0, ref_vuv):
    assert len(f0) == len(ref_f0)
    assert np.array_equal(vuv, ref_vuv)
    v = ref_vuv > 0
    rel = np.abs(f0[v] - ref_f0[v]) / ref_f0[v]
    step = 2 ** (1 / 768) - 1
    exact = rel < 1e-12
    one_step = np.abs(rel - step) < 1e-6
    assert np.all(exact | one_step), float(rel.max())
    assert one_step.sum() <= 2, int(one_step.sum())
    assert np.all(f0[~v] == 0)


@pytest.mark.parametrize("tag", ["16k", "48k", "22k"])
def test_swipe_vs_reference(golden, tag):
    from world._synthetic import synth_utterance
    from world.swipe import swipe

    g = golden("swipe")
    fs, u, sec = g["args_" + tag]
    x = synth_utterance(int(u), int(fs), float(sec))
    r = swipe(int(fs), x, [71, 800], 0.005, 0.3)
    assert np.array_equal(r["temporal_positions"], g["tp_" + tag])
    _compare(r["f0"], r["vuv"], g["f0_" + tag], g["vuv_" + tag])
    if tag == "16k":
        # no threshold: every frame gets a pitch.  On the frames the 0.3 threshold calls voiced the maximum is
        # pronounced and must agree; on the others (noise: a flat strength profile whose argmax hops between distant
        # candidates at the 1e-13 level) only the range is checked
        r2 = swipe(int(fs), x, [71, 800], 0.005)
        strong = g["vuv_16k"] > 0
        assert np.all(r2["f0"] > 0) and np.all(g["f0_16k_nothr"] > 0)
        _compare(r2["f0"][strong], np.ones(strong.sum()), g["f0_16k_nothr"][strong], np.ones(strong.sum()))
        assert np.all((r2["f0"] >= 71 * (1 - 1e-9)) & (r2["f0"] <= 800))
        assert np.mean(np.abs(r2["f0"] / g["f0_16k_nothr"] - 1) < 1e-12) > 0.7


def test_swipe_on_sieve_quirk_candidates(golden):
    from world._synthetic import harmonic_tone
    from world.swipe import swipe

    g = golden("swipe")
    assert len(g["tone_cases"]) == 9
    for fs, f0 in g["tone_cases"]:
        r = swipe(int(fs), harmonic_tone(int(fs), float(f0)), [71, 800], 0.005, 0.3)
        _compare(r["f0"], r["vuv"], g["tone_f0_%d_%d" % (fs, f0)], g["tone_vuv_%d_%d" % (fs, f0)])


def test_swipe_test_wav_and_facade(golden):
    from scipy.io import wavfile

    from world import main
    from world._synthetic import synth_utterance
    from world.swipe import swipe

    g = golden("swipe")
    fs, xi = wavfile.read(os.path.join(GOLDEN, "test-mwm.wav"))
    r = swipe(fs, xi / (2 ** 15 - 1), [71, 800], 0.005, 0.3)
    _compare(r["f0"], r["vuv"], g["f0_mwm"], g["vuv_mwm"])
    x = synth_utterance(0, 16000, 1.2)
    dat = main.World().encode(16000, x, f0_method="swipe")
    _compare(dat["f0"], dat["vuv"], g["enc_f0"], g["enc_vuv"])
    same = np.abs(dat["f0"] - g["enc_f0"]) < 1e-9  # spectra depend on f0: compare the frames whose f0 agrees exactly
    assert np.max(np.abs(dat["spectrogram"].sum(axis=0)[same] / g["enc_spec_colsum"][same] - 1)) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=0)[same] - g["enc_ap_colsum"][same])) < 1e-5


def test_swipe_batch_equals_single():
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.swipe import swipe

    fs = 16000
    xs = [synth_utterance(70 + i, fs, 0.5 + 0.3 * i) for i in range(3)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="swipe")
    fo = enc.batch.frame_off
    f0 = enc.f0.cpu().numpy()
    for u, x in enumerate(xs):
        one = swipe(fs, x, [71, 800], 0.005, 0.3)
        got = f0[int(fo[u]):int(fo[u + 1])]
        assert np.array_equal(got > 0, one["f0"] > 0)
        assert np.array_equal(got[got > 0], one["f0"][one["f0"] > 0])


@pytest.mark.parametrize("fs,floor", [(96000, 71), (88200, 71), (44100, 40.9), (48000, 45), (96000, 57.1), (88200, 50)])
def test_swipe_with_8192_and_16384_sample_windows(fs, floor):
    """The longest window is 2^round(log2(8 fs / f0_floor)) samples (world/swipe.py:33-35): 8192 from 88.2 kHz up at the
    default floor and for floors below ~60 Hz at 44.1 / 48 kHz, 16384 below ~66 Hz at 88.2 / 96 kHz (round 6: refused before —
    found by the differential campaign)."""
    from oracle import pitch_swipe
    from world._synthetic import synth_utterance
    from world.swipe import swipe

    x = synth_utterance(7, fs, 0.5)
    o = pitch_swipe.swipe_np(fs, x, [floor, 800], sTHR=0.3)
    d = swipe(fs, x, [floor, 800], 0.005, 0.3)
    assert np.array_equal(d["vuv"], o["vuv"]) and o["vuv"].sum() > 50
    v = o["vuv"] > 0
    assert np.max(np.abs(d["f0"][v] - o["f0"][v]) / o["f0"][v]) < 1e-12
