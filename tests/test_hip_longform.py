"""GPU: BASELINE config 5's defining property — a 60 s utterance at 48 kHz through Harvest, CheapTrick (fft 2048),
D4C (fft 4096), scale_pitch(1.5) + scale_duration(2.0) and the pulse-wise decode (5.76 M output samples: the exact
phase accumulator at scale).  The oracle is too slow for 60 s of analysis, so:
  * Harvest: the whole 60 s contour against a fixture generated from the reference itself;
  * CheapTrick / D4C parity on a 5 s interior window (they are frame-local: same frames, same numbers);
  * batch == single bitwise for the long utterance next to a short one;
  * decode: exact output length (Q9), exact pulse count and noise-draw count against the oracle's np.cumsum pulse
    train over all 5.76 M samples, waveform parity on the first 10 s against the oracle decode of the truncated
    parameter tracks (identical noise stream)."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu

FS = 48000
SECONDS = 60.0


@pytest.fixture(scope="module")
def longform():
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    from conftest import synth_cached
    x = synth_cached(75, FS, SECONDS)
    short = synth_utterance(76, FS, 0.7)
    wb = WorldBatch()
    enc = wb.encode([x], FS, f0_method="harvest")
    return {"x": x, "short": short, "wb": wb, "enc": enc, "dict": enc.to_dicts()[0]}


def test_longform_shapes_and_batch_equals_single(longform):
    wb, enc, d = longform["wb"], longform["enc"], longform["dict"]
    nf = int(1000 * len(longform["x"]) / FS / 5 + 1)
    assert nf == 12001 and len(d["f0"]) == nf
    assert d["spectrogram"].shape == (1025, nf) and d["aperiodicity"].shape == (1025, nf)
    assert np.all(np.isfinite(d["spectrogram"])) and np.all(np.isfinite(d["aperiodicity"]))
    voiced = d["vuv"] > 0
    assert 0.6 < voiced.mean() < 0.9  # the generator gates 0.8 s voiced / 0.2 s unvoiced
    enc2 = wb.encode([longform["short"], longform["x"]], FS, f0_method="harvest")
    d2 = enc2.to_dicts()[1]
    for key in ("f0", "vuv", "spectrogram", "aperiodicity", "temporal_positions"):
        assert np.array_equal(d[key], d2[key]), key


def test_longform_harvest_vs_reference_fixture(longform, golden):
    """Harvest over the whole 60 s (2 880 000 samples, 60 001 1 ms frames) against the reference's own output
    (tests/golden/golden_longform48k.npz, make_golden.py longform_fixture)."""
    g = golden("longform48k")
    d = longform["dict"]
    assert int(g["utt"]) == 75 and float(g["seconds"]) == SECONDS
    assert np.array_equal(d["temporal_positions"], g["tp"])
    assert np.array_equal(d["vuv"], g["harvest_vuv"])
    # encode() zeroes f0 on unvoiced frames (d4c.py:32); Harvest's own output is already zero there
    assert np.max(np.abs(d["f0"] - g["harvest_f0"])) < 1e-6


def test_longform_analysis_window_vs_oracle(longform):
    from oracle import aperiodicity, envelope

    x, d = longform["x"], longform["dict"]
    f_lo, n_f = 4000, 1001                     # frames 20 s .. 25 s
    a = f_lo * 240                             # 5 ms hop = 240 samples: the window starts on a frame centre
    xw = x[a:a + (n_f - 1) * 240 + 1]
    sl = slice(100, n_f - 100)                 # interior: clear of the window edges (longest analysis window 4/47 s)
    f0w, vuvw = d["f0"][f_lo:f_lo + n_f], d["vuv"][f_lo:f_lo + n_f]
    tpw = np.arange(n_f) * 0.005
    # CheapTrick / D4C with the GPU's f0 contour: frame-local, so interior frames agree to FP64 round-off
    spec, _, _ = envelope.cheaptrick_np(xw, FS, f0w, vuvw, tpw, want_ps=False)
    g = d["spectrogram"][:, f_lo:f_lo + n_f]
    assert rel_rms(g[:, sl], spec[:, sl]) < 1e-8
    ap, _, _ = aperiodicity.d4c_np(xw, FS, f0w, vuvw, tpw)
    ga = d["aperiodicity"][:, f_lo:f_lo + n_f]
    assert np.max(np.abs(ga[:, sl] - ap[:, sl])) < 1e-6
    assert rel_rms(ga[:, sl], ap[:, sl]) < 1e-8


def test_longform_modifiers_and_decode(longform):
    from oracle import resynth
    from world.synthesis import synthesis_device, synthesis_plan, time_axis_params

    wb, enc = longform["wb"], longform["enc"]
    enc.scale_pitch(1.5).scale_duration(2.0)
    d = enc.to_dicts()[0]
    ny, t0, dt = time_axis_params(d["temporal_positions"], FS)
    assert ny == len(np.arange(0.0, 2 * 60.0 + 1 / FS, 1 / FS))  # NumPy float-arange length (Q9)
    # the whole 5.76 M-sample phase accumulation: pulse count and reference randn draw count are exact
    times, idx, _, _, _ = resynth.pulse_train(d["temporal_positions"], d["f0"], FS, d["vuv"])
    counts, draws = synthesis_plan(wb.rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, FS, [ny], [t0], [dt],
                                   ny // 2 + 16)
    assert int(counts[0]) == len(idx)
    ns = np.diff(np.r_[idx, idx[-1]])
    assert int(draws[0]) == int(np.sum(np.maximum(3, ns)))
    rng = np.random.RandomState(11)
    noise = [rng.randn(int(draws[0]))]
    y, y_off = wb.decode_device(enc, noise=noise)
    assert y_off[1] == ny
    y = y.cpu().numpy()
    assert np.all(np.isfinite(y)) and np.max(np.abs(y)) <= 1.0 + 1e-12
    # the same synthesis without decode()'s peak normalisation (world/main.py:209-212), to compare raw samples
    y_raw, _ = synthesis_device(wb.rt, enc.batch, enc.temporal_positions, enc.f0, enc.vuv, enc.spectrogram,
                                enc.aperiodicity, FS, enc.fft_size, [ny], [t0], [dt], noise_d=wb.rt.to_device(noise[0]),
                                noise_off=[0, len(noise[0])])
    y_raw = y_raw.cpu().numpy()
    peak = max(1.0, float(np.max(np.abs(y_raw))))
    assert rel_rms(y, y_raw / peak) < 1e-12
    # waveform parity on the first 10 s: oracle decode of the first 1001 frames with the same noise stream
    nt = 1001
    yo = resynth.synthesis_np(d["f0"][:nt], d["vuv"][:nt], d["temporal_positions"][:nt], d["spectrogram"][:, :nt],
                              d["aperiodicity"][:, :nt], FS, noise=noise[0])
    cmp_n = len(yo) - 3 * 2048  # clear of the truncation (the last pulses before the cut see a different successor)
    assert rel_rms(y_raw[:cmp_n], yo[:cmp_n]) < 1e-8
    assert wb.rt.take_flags() == [0] * 16
