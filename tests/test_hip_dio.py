"""GPU parity: wh_dio / wh_stonemask vs the golden fixtures (reference output) and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_dio_vs_golden(golden, tag):
    from world.dio import dio

    g = golden(tag)
    fs = int(g["fs"])
    d = dio(g["x"], fs, _index_bias=g["dio_index_bias"])
    assert len(d["f0"]) == len(g["dio_f0"])          # frame count is bit-exact
    assert np.array_equal(d["temporal_positions"], g["tp"])
    # the reference filters by FFT, the kernel by direct FIR: candidates agree to ~1e-9 Hz
    assert np.max(np.abs(d["raw_f0_candidates"] - g["dio_raw"])) < 1e-6
    assert np.max(np.abs(d["f0_candidates"] - g["dio_cands"])) < 1e-6
    assert np.array_equal(d["vuv"], g["dio_vuv"])
    assert np.max(np.abs(d["f0"] - g["dio_f0"])) < 1e-6


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_stonemask_vs_golden(golden, tag):
    from world.stonemask import stonemask

    g = golden(tag)
    fs = int(g["fs"])
    f0_in = g["dio_f0"].copy()
    out = stonemask(g["x"], fs, g["tp"], f0_in)
    assert np.array_equal(f0_in, g["dio_f0"])  # input untouched
    assert np.array_equal(out == 0, g["stonemask_f0"] == 0)
    voiced = g["stonemask_f0"] != 0
    assert np.max(np.abs(out[voiced] - g["stonemask_f0"][voiced]) / g["stonemask_f0"][voiced]) < 1e-9


def test_dio_stonemask_mwm_config1(golden):
    """22.05 kHz test-mwm.wav (decimation ratio 5, Q4) against the reference's own DIO+StoneMask f0."""
    from scipy.io import wavfile

    from world.dio import dio
    from world.stonemask import stonemask

    g = golden("mwm")
    fs, xi = wavfile.read(os.path.join(os.path.dirname(__file__), "golden", "test-mwm.wav"))
    x = xi / (2 ** 15 - 1)
    d = dio(x, fs)
    f0 = stonemask(x, fs, d["temporal_positions"], d["f0"])
    assert np.array_equal(d["vuv"], g["dio_vuv"])
    # golden dio_f0 went through cheaptrick/d4c bookkeeping: zero where vuv==0, 500 Hz never survives there
    ref = g["dio_f0"]
    voiced = g["dio_vuv"] != 0
    assert np.max(np.abs(f0[voiced] - ref[voiced]) / ref[voiced]) < 1e-8


def test_dio_ragged_batch_matches_single():
    """Two utterances of different length in one batch == each one alone (no cross-utterance state)."""
    from world import _hip, _tables
    from world._synthetic import synth_utterance
    from world.dio import dio, dio_device

    fs = 16000
    xs = [synth_utterance(11, fs, 0.7), synth_utterance(12, fs, 1.1)]
    rt = _hip.Runtime.get()
    nfs = [_tables.frame_count(len(x), fs, 5) for x in xs]
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    f_off = np.concatenate([[0], np.cumsum(nfs)])
    batch = rt.make_batch(x_off, f_off)
    tp = np.concatenate([_tables.frame_times(n, 5) for n in nfs])
    f0, vuv, _, _ = dio_device(rt, batch, rt.to_device(np.concatenate(xs)), rt.to_device(tp), fs)
    f0 = f0.cpu().numpy()
    for u, x in enumerate(xs):
        single = dio(x, fs)
        assert np.array_equal(f0[f_off[u]:f_off[u + 1]], single["f0"])
    assert rt.take_flags() == [0] * 16


def test_dio_long_utterance_matches_oracle():
    """15 s at 16 kHz (3001 frames): the contour walk's candidate rows no longer fit the CU's LDS and take the global
    path (contour_kernel), the band walker runs more than one segment per band; a 1.3 s utterance in the same batch
    takes the LDS path.  Voicing exact, f0 to 1e-6 Hz against the NumPy oracle."""
    from oracle import pitch_dio
    from world import _hip, _tables
    from world._synthetic import synth_utterance
    from world.dio import dio_device

    fs = 16000
    xs = [synth_utterance(31, fs, 15.0), synth_utterance(32, fs, 1.3)]
    rt = _hip.Runtime.get()
    nfs = [_tables.frame_count(len(x), fs, 5) for x in xs]
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
    f_off = np.concatenate([[0], np.cumsum(nfs)])
    batch = rt.make_batch(x_off, f_off)
    tp = np.concatenate([_tables.frame_times(n, 5) for n in nfs])
    f0, vuv, _, _ = dio_device(rt, batch, rt.to_device(np.concatenate(xs)), rt.to_device(tp), fs)
    f0, vuv = f0.cpu().numpy(), vuv.cpu().numpy()
    for u, x in enumerate(xs):
        o = pitch_dio.dio_np(x, fs)
        sl = slice(f_off[u], f_off[u + 1])
        assert np.array_equal(vuv[sl], o["vuv"])
        assert np.max(np.abs(f0[sl] - o["f0"])) < 1e-6
    assert rt.take_flags() == [0] * 16


@pytest.mark.parametrize("kw", [dict(f0_floor=50, f0_ceil=500), dict(channels_in_octave=4, frame_period=10),
                                dict(f0_floor=100, f0_ceil=1000, allowed_range=0.2, frame_period=2)])
def test_dio_other_parameters(kw):
    """Non-default search range / band density / hop: band count, tap lengths (odd and even) and the contour's
    minimum voiced run all follow from them."""
    from oracle import pitch_dio
    from world._synthetic import synth_utterance
    from world.dio import dio

    fs = 16000
    x = synth_utterance(42, fs, 1.7)
    o = pitch_dio.dio_np(x, fs, **kw)
    d = dio(x, fs, **kw)
    assert np.array_equal(d["temporal_positions"], o["temporal_positions"])
    assert np.array_equal(d["vuv"], o["vuv"])
    assert np.max(np.abs(d["f0"] - o["f0"])) < 1e-6


@pytest.mark.parametrize("fs", [16000, 48000])
def test_stonemask_mixed_time_grids(fs):
    """Frame times on and off the sample grid (the tabulated kernel evaluates the sample picks per tap like the
    reference; its windows do not depend on the frame time) and windows that reach before the signal start (left to
    the staged kernel), frame by frame within one call: against the oracle."""
    from oracle import pitch_dio
    from world._synthetic import synth_utterance
    from world.stonemask import stonemask

    x = synth_utterance(43, fs, 1.4)
    d = pitch_dio.dio_np(x, fs)
    tp = d["temporal_positions"].copy()
    f0 = d["f0"].copy()
    f0[:4] = 180.0                      # voiced frames whose windows reach before sample 1
    tp[1::3] += 0.37 / fs               # every third frame between two samples
    tp[2::3] *= 1.0 + 1e-13             # ... and one within rounding of the grid
    ref = pitch_dio.stonemask_np(x, fs, tp, f0)
    out = stonemask(x, fs, tp, f0)
    assert np.array_equal(out == 0, ref == 0)
    v = ref != 0
    assert np.max(np.abs(out[v] - ref[v]) / ref[v]) < 1e-9
