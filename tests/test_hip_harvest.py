"""GPU parity: wh_harvest vs the golden fixtures (reference output) and the oracle's intermediates."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_harvest_stages_vs_oracle(golden, tag):
    from oracle import pitch_harvest
    from world import _hip, _tables
    from world.harvest import harvest_device

    g = golden(tag)
    fs = int(g["fs"])
    x = g["x"]
    o = pitch_harvest.harvest_np(x, fs, return_aux=True)
    aux = o["aux"]
    rt = _hip.Runtime.get()
    nf = _tables.frame_count(len(x), fs, 5)
    tp = _tables.frame_times(nf, 5)
    batch = rt.make_batch([0, len(x)], [0, nf])
    f0, vuv, dbg = harvest_device(rt, batch, rt.to_device(x), rt.to_device(tp), fs, debug=True)
    y = dbg["y"].cpu().numpy()[: len(aux["y"])]
    assert np.max(np.abs(y - aux["y"])) < 1e-12          # decimation (SciPy filtfilt semantics)
    nb = aux["raw"].shape[0]
    raw = dbg["raw"].cpu().numpy()[: nb * aux["raw"].shape[1]].reshape(nb, -1)
    live_mismatch = np.sum((raw != 0) != (aux["raw"] != 0))
    assert live_mismatch == 0
    assert np.max(np.abs(raw - aux["raw"])) < 1e-6       # Hz; FFT-conv vs direct FIR
    f1 = dbg["f0_1ms"].cpu().numpy()[: len(aux["f0_1ms"])]
    assert np.array_equal(f1 != 0, aux["f0_1ms"] != 0)
    assert np.max(np.abs(f1 - aux["f0_1ms"])) < 1e-6
    assert np.array_equal(vuv.cpu().numpy(), g["harvest_vuv"])
    assert np.max(np.abs(f0.cpu().numpy() - g["harvest_f0"])) < 1e-6
    assert rt.take_flags() == [0] * 16


def test_harvest_mwm_config1(golden):
    """BASELINE config 1 input: test-mwm.wav (22.05 kHz → decimation ratio 3) against the reference's f0."""
    from scipy.io import wavfile

    from world.harvest import harvest

    g = golden("mwm")
    fs, xi = wavfile.read(os.path.join(os.path.dirname(__file__), "golden", "test-mwm.wav"))
    x = xi / (2 ** 15 - 1)
    h = harvest(x, fs)
    assert np.array_equal(h["temporal_positions"], g["tp"])
    assert np.array_equal(h["vuv"], g["vuv"])
    ref = g["f0"]  # after cheaptrick/d4c bookkeeping: 0 where unvoiced
    voiced = g["vuv"] != 0
    assert np.max(np.abs(h["f0"][voiced] - ref[voiced]) / ref[voiced]) < 1e-8


@pytest.mark.parametrize("floor,ceil,period", [(40, 600, 5), (90, 400, 10), (71, 800, 2), (20, 400, 5)])
def test_harvest_other_search_ranges(floor, ceil, period):
    """Non-default f0 range / frame period: the longest refinement window, the band set (and with a 40 Hz floor the
    direct-FIR fallback of the band filters: taps longer than an overlap-save tile allows) all change with them; at a
    20 Hz floor the refinement's transform tables no longer fit LDS (global twiddles, rotation windows)."""
    from oracle import pitch_harvest
    from world._synthetic import synth_utterance
    from world.harvest import harvest

    fs = 16000
    x = synth_utterance(41, fs, 1.6)
    o = pitch_harvest.harvest_np(x, fs, f0_floor=floor, f0_ceil=ceil, frame_period=period)
    d = harvest(x, fs, f0_floor=floor, f0_ceil=ceil, frame_period=period)
    assert np.array_equal(d["temporal_positions"], o["temporal_positions"])
    assert np.array_equal(d["vuv"], o["vuv"])
    assert np.max(np.abs(d["f0"] - o["f0"])) < 1e-6
