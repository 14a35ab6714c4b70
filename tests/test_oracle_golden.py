"""Pin the NumPy oracle (oracle/) against the fixtures generated from the real reference
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest

from conftest import rel_rms
from oracle import aperiodicity, api, envelope, pitch_dio, pitch_harvest, resynth

TAGS = ["syn16k", "syn48k"]


@pytest.mark.parametrize("tag", TAGS)
def test_dio_stonemask(golden, tag):
    g = golden(tag)
    fs = int(g["fs"])
    d = pitch_dio.dio_np(g["x"], fs)
    assert [t[1] for t in pitch_dio.dio_band_tables(71 * 2.0 ** ((np.arange(len(g["dio_index_bias"])) + 1) / 2), 4000)] \
        == list(g["dio_index_bias"]), "host libm/BLAS picks a different Nuttall argmax than the fixture (SURVEY Q5)"
    assert np.array_equal(d["vuv"], g["dio_vuv"])
    assert np.allclose(d["raw_f0_candidates"], g["dio_raw"], rtol=0, atol=1e-8)
    assert np.allclose(d["f0_candidates"], g["dio_cands"], rtol=0, atol=1e-8)
    assert np.allclose(d["f0"], g["dio_f0"], rtol=0, atol=1e-8)
    sm = pitch_dio.stonemask_np(g["x"], fs, g["tp"], g["dio_f0"])
    assert np.allclose(sm, g["stonemask_f0"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("tag", TAGS)
def test_cheaptrick(golden, tag):
    g = golden(tag)
    fs = int(g["fs"])
    sp, ps, f0u = envelope.cheaptrick_np(g["x"], fs, g["stonemask_f0"], g["dio_vuv"], g["tp"])
    assert sp.shape == g["ct_spectrogram"].shape
    assert np.array_equal(f0u, g["ct_f0_after"])
    assert rel_rms(sp, g["ct_spectrogram"]) < 1e-10
    # eps dither of the reference (Q10) only matters on ~1e-16-level bins
    assert np.max(np.abs(sp - g["ct_spectrogram"]) / g["ct_spectrogram"]) < 1e-6
    assert np.allclose(ps[:, g["ct_ps_cols"]], g["ct_ps"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", TAGS)
def test_d4c_and_requiem(golden, tag):
    g = golden(tag)
    fs = int(g["fs"])
    ap, coarse, f0o = aperiodicity.d4c_np(g["x"], fs, g["ct_f0_after"], g["dio_vuv"], g["tp"])
    assert np.array_equal(f0o, g["d4c_f0_after"])
    assert np.allclose(coarse, g["d4c_coarse"], rtol=0, atol=1e-8)
    assert np.allclose(ap, g["d4c_aperiodicity"], rtol=0, atol=1e-9)
    band, _ = aperiodicity.d4c_requiem_np(g["x"], fs, g["ct_f0_after"], g["dio_vuv"], g["tp"])
    assert band.shape == g["req_band_ap"].shape
    assert np.allclose(band, g["req_band_ap"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("tag", TAGS)
def test_synthesis(golden, tag):
    g = golden(tag)
    fs = int(g["fs"])
    np.random.seed(int(g["seed"]))
    y = resynth.synthesis_np(g["d4c_f0_after"], g["dio_vuv"], g["tp"], g["ct_spectrogram"], g["d4c_aperiodicity"], fs)
    assert len(y) == len(g["syn_y"])
    assert np.allclose(y, g["syn_y"], rtol=0, atol=1e-12)
    # modifiers: scale_pitch(1.5) + scale_duration(2.0) (world/main.py:154-178)
    np.random.seed(int(g["seed"]) + 1)
    y2 = resynth.synthesis_np(g["d4c_f0_after"] * 1.5, g["dio_vuv"], g["tp"] * 2.0, g["ct_spectrogram"],
                              g["d4c_aperiodicity"], fs)
    assert len(y2) == int(g["mod_len"])
    assert np.allclose(y2[:2048], g["mod_head"], rtol=0, atol=1e-12)
    assert np.allclose(y2[-2048:], g["mod_tail"], rtol=0, atol=1e-12)
    assert np.allclose(np.add.reduceat(y2, np.arange(0, len(y2), 256)), g["mod_blocksum"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("tag", TAGS)
def test_requiem_synthesis_and_seeds(golden, tag):
    g = golden(tag)
    fs = int(g["fs"])
    random.seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    seeds = resynth.seeds_np(fs)
    assert np.allclose(seeds["pulse"], g["seeds_pulse"], rtol=0, atol=1e-15)
    assert np.allclose(seeds["noise"], g["seeds_noise"], rtol=0, atol=1e-13)
    f0r = np.where(g["dio_vuv"] == 0, 0.0, g["ct_f0_after"])
    y, cur = resynth.synthesis_requiem_np(f0r, g["dio_vuv"], g["tp"], g["ct_spectrogram"], g["req_band_ap"], fs,
                                          {"pulse": g["seeds_pulse"], "noise": g["seeds_noise"]})
    assert np.allclose(y, g["req_y"], rtol=0, atol=1e-12)
    assert np.array_equal(cur, g["req_cursor"])


@pytest.mark.parametrize("tag", TAGS)
def test_harvest(golden, tag):
    g = golden(tag)
    h = pitch_harvest.harvest_np(g["x"], int(g["fs"]))
    assert np.array_equal(h["vuv"], g["harvest_vuv"])
    assert np.allclose(h["f0"], g["harvest_f0"], rtol=1e-10, atol=0)


def test_tables(golden):
    from oracle import common

    g = golden("tables")
    for fs, n in ((16000, 160000), (22050, 102400), (48000, 480000)):
        nf = common.frame_count(n, fs, 5)
        assert nf == int(g["F_%d" % fs])
        tp = common.frame_times(nf, 5)
        assert resynth.output_length(tp, fs) == int(g["Ny_%d" % fs])
        assert resynth.output_length(tp * 2.0, fs) == int(g["Ny2_%d" % fs])
        assert envelope.default_fft_size(fs) == int(g["ct_fft_%d" % fs])


def test_config1_mwm_end_to_end(golden):
    """BASELINE config 1 through the oracle facade: encode(harvest) + decode on test-mwm.wav."""
    import os

    from scipy.io import wavfile

    g = golden("mwm")
    fs, xi = wavfile.read(os.path.join(os.path.dirname(__file__), "golden", "test-mwm.wav"))
    x = xi / (2 ** 15 - 1)
    dat = api.encode_np(fs, x, f0_method="harvest")
    assert np.array_equal(dat["vuv"], g["vuv"])
    assert np.allclose(dat["f0"], g["f0"], rtol=1e-9, atol=0)
    assert np.allclose(dat["spectrogram"].sum(axis=0), g["spec_colsum"], rtol=1e-8)
    # the reference's eps dither (Q10) is visible on ~1e-11-level bins: bound per-bin loosely, RMS tightly
    assert np.allclose(dat["spectrogram"][:, g["cols"]], g["spec_cols"], rtol=1e-4, atol=1e-300)
    assert rel_rms(dat["spectrogram"][:, g["cols"]], g["spec_cols"]) < 1e-10
    assert np.allclose(dat["aperiodicity"].sum(axis=1), g["ap_rowsum"], rtol=1e-8)
    assert np.allclose(dat["aperiodicity"][:, g["cols"]], g["ap_cols"], rtol=0, atol=1e-8)
    np.random.seed(int(g["seed"]))
    dat = api.decode_np(dat)
    y = dat["out"]
    assert len(y) == int(g["out_len_std"])
    # chained encode→decode: the reference's unseeded eps dither in CheapTrick (Q10) propagates at ~1e-8
    assert np.allclose(y[:4096], g["out_head_std"], rtol=0, atol=1e-7)
    assert np.allclose(y[-4096:], g["out_tail_std"], rtol=0, atol=1e-7)
    assert np.allclose(np.add.reduceat(y, np.arange(0, len(y), 256)), g["out_blocksum_std"], rtol=0, atol=1e-6)


def test_getters_fixture_pins_oracle(golden):
    """A-3 fixture (World.get_f0 / get_spectrum / encode_w_gvn_f0 of the reference) against the oracle stages."""
    from world._synthetic import synth_utterance

    g = golden("getters")
    fs = int(g["fs"])
    x = synth_utterance(int(g["utt"]), fs, float(g["seconds"]))
    h = pitch_harvest.harvest_np(x, fs)
    assert np.array_equal(h["vuv"], g["getf0_harvest_vuv"]) and np.array_equal(h["f0"], g["getf0_harvest_f0"])
    d = pitch_dio.dio_np(x, fs)
    f0 = pitch_dio.stonemask_np(x, fs, d["temporal_positions"], d["f0"])
    assert np.array_equal(d["vuv"], g["getf0_dio_vuv"])
    assert np.allclose(f0, g["getf0_dio_f0"], rtol=1e-12, atol=0)
    sp, ps, f0u = envelope.cheaptrick_np(x, fs, f0, d["vuv"], d["temporal_positions"])
    assert np.allclose(f0u, g["getspec_f0"], rtol=1e-12, atol=0)
    assert rel_rms(sp.sum(axis=0), g["getspec_colsum"]) < 1e-10
    assert rel_rms(sp[:16, :16], g["getspec_head"]) < 1e-10
    sp2, _, f0u2 = envelope.cheaptrick_np(x, fs, g["gvn_src_f0"], g["gvn_src_vuv"], g["gvn_src_tp"], fft_size=1024)
    ap, coarse, f0o = aperiodicity.d4c_np(x, fs, f0u2, g["gvn_src_vuv"], g["gvn_src_tp"], fft_size_for_spectrum=1024)
    assert np.allclose(f0o, g["gvn_f0"], rtol=1e-12, atol=0)
    assert rel_rms(sp2.sum(axis=0), g["gvn_spec_colsum"]) < 1e-10
    assert rel_rms(ap.sum(axis=0), g["gvn_ap_colsum"]) < 1e-9
    assert np.max(np.abs(coarse - g["gvn_coarse"])) < 1e-8


@pytest.mark.parametrize("tag", ["16k", "48k", "22k"])
def test_swipe_oracle_vs_reference(golden, tag):
    """oracle/pitch_swipe.py against the reference's swipe() output (world/swipe.py:9-105)."""
    from oracle import pitch_swipe
    from world._synthetic import synth_utterance

    g = golden("swipe")
    fs, u, sec = g["args_" + tag]
    r = pitch_swipe.swipe_np(int(fs), synth_utterance(int(u), int(fs), float(sec)), [71, 800], 0.005, 0.3)
    assert np.array_equal(r["vuv"], g["vuv_" + tag])
    assert np.array_equal(r["f0"], g["f0_" + tag])


def _sieve_fixture(g):
    off = g["sieve_off"]
    return [list(map(int, g["sieve_flat"][off[n]:off[n + 1]])) for n in range(len(off) - 1)]


def test_sieve_quirk_is_restated(golden):
    """world/swipe.py:158-172 keeps n when n is the square of a prime; the oracle's and the product's harmonic lists
    must be the reference's for every n the candidate kernels can ask for (fixture: the reference's sieve(0..400))."""
    from oracle.pitch_swipe import sieve_as_reference
    from world.swipe import _sieve_harmonics

    ref = _sieve_fixture(golden("swipe"))
    assert ref[9] == [2, 3, 5, 7, 9] and ref[4] == [2, 3, 4] and ref[25][-1] == 25 and ref[27][-1] == 23
    for n, want in enumerate(ref):
        assert sieve_as_reference(n) == want, n
        assert _sieve_harmonics(n) == want, n


def test_swipe_oracle_on_quirk_candidates(golden):
    """Tones sweeping over the candidates whose kernel holds a prime square (157-158, 297-305, 737-798 Hz at 16 kHz;
    their counterparts at 22.05 and 48 kHz): the oracle must give the reference's f0 on every frame, bit for bit.
    (With true primes instead of the reference's sieve up to 42 of 161 frames move by a grid step.)"""
    from oracle import pitch_swipe
    from world._synthetic import harmonic_tone

    g = golden("swipe")
    for fs, f0 in g["tone_cases"]:
        r = pitch_swipe.swipe_np(int(fs), harmonic_tone(int(fs), float(f0)), [71, 800], 0.005, 0.3)
        assert np.array_equal(r["f0"], g["tone_f0_%d_%d" % (fs, f0)]), (fs, f0)
        assert np.array_equal(r["vuv"], g["tone_vuv_%d_%d" % (fs, f0)])


def test_hires_96k_chain(golden):
    """96 kHz (8192-point D4C transforms): the oracle against the reference's harvest -> cheaptrick -> d4c / d4cRequiem
    -> seeded synthesis (make_golden.py hires_fixture)."""
    g = golden("syn96k")
    fs = int(g["fs"])
    h = pitch_harvest.harvest_np(g["x"], fs)
    assert np.array_equal(h["vuv"], g["harvest_vuv"]) and np.array_equal(h["f0"], g["harvest_f0"])
    sp, _, f0u = envelope.cheaptrick_np(g["x"], fs, g["harvest_f0"], g["harvest_vuv"], g["tp"])
    assert np.array_equal(f0u, g["ct_f0_after"]) and rel_rms(sp, g["ct_spectrogram"]) < 1e-10
    ap, coarse, f0o = aperiodicity.d4c_np(g["x"], fs, g["ct_f0_after"], g["harvest_vuv"], g["tp"])
    assert np.array_equal(f0o, g["d4c_f0_after"])
    assert np.allclose(coarse, g["d4c_coarse"], rtol=0, atol=1e-8)
    assert np.allclose(ap, g["d4c_aperiodicity"], rtol=0, atol=1e-9)
    band, _ = aperiodicity.d4c_requiem_np(g["x"], fs, g["ct_f0_after"], g["harvest_vuv"], g["tp"])
    assert np.allclose(band, g["req_band_ap"], rtol=0, atol=1e-8)
    np.random.seed(int(g["seed"]))
    y = resynth.synthesis_np(g["d4c_f0_after"], g["harvest_vuv"], g["tp"], g["ct_spectrogram"], g["d4c_aperiodicity"], fs)
    assert len(y) == len(g["syn_y"]) and np.max(np.abs(y - g["syn_y"])) < 1e-10
