"""Oracle faithfulness over World.encode's arguments (CPU): tests/golden/golden_sweep.npz is written by `make_golden.py
sweep`, which runs the UNMODIFIED reference's World.encode + seeded decode next to oracle/api.py on the twelve cases of
tests/_sweep_cases.py — search ranges, frame periods of 1 / 2.5 / 3 / 10 / 12.5 ms, DIO's channels / target rate / allowed
range, the fft_size override, 8 / 11.025 / 24 / 32 / 44.1 kHz, an int16-scaled waveform, a length on a filter-tile edge.
The worst reference-vs-oracle error per tensor as measured there, and the oracle re-run here against the reference's
f0 / vuv / frame times / tensor sums / decoded block sums without the reference."""
import random

import numpy as np
import pytest

from _sweep_cases import sweep_cases, sweep_input
from conftest import rel_rms
from oracle import api

CASES = sweep_cases()


def test_recorded_worst_errors(golden):
    g = golden("sweep")
    assert len(g["err_vuv_mismatch"]) == len(CASES) == 12
    for k in ("vuv_mismatch", "frames_mismatch", "out_len_mismatch", "tp_maxabs"):
        assert float(g["worst_" + k]) == 0.0, k
    assert float(g["worst_f0_maxrel"]) < 1e-12
    assert float(g["worst_spectrogram_relrms"]) < 1e-12   # the reference's rand*eps dither vs eps/2
    assert float(g["worst_aperiodicity_maxabs"]) < 1e-10
    assert float(g["worst_out_relrms"]) < 1e-10


@pytest.mark.parametrize("i", range(len(CASES)))
def test_oracle_reproduces_reference_case(golden, i):
    from world._synthetic import synth_utterance

    g = golden("sweep")
    _, fs, _, _, kw = CASES[i]
    x = sweep_input(synth_utterance, CASES[i])
    dat = api.encode_np(fs, x, **kw)
    assert np.array_equal(dat["temporal_positions"], g["tp_%d" % i])
    assert np.array_equal(dat["vuv"], g["vuv_%d" % i])
    assert np.allclose(dat["f0"], g["f0_%d" % i], rtol=1e-12, atol=0)
    assert list(dat["spectrogram"].shape) == list(g["spec_shape_%d" % i])
    assert rel_rms(dat["spectrogram"].sum(axis=0), g["spec_colsum_%d" % i]) < 1e-12
    assert rel_rms(dat["spectrogram"].sum(axis=1), g["spec_rowsum_%d" % i]) < 1e-12
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=0) - g["ap_colsum_%d" % i])) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=1) - g["ap_rowsum_%d" % i])) < 1e-8
    random.seed(int(g["seed"]) + 200 + i)
    np.random.seed(int(g["seed"]) + 200 + i)
    y = api.decode_np(dat)["out"]
    assert len(y) == int(g["out_len_%d" % i])
    assert np.max(np.abs(np.add.reduceat(y, np.arange(0, len(y), 256)) - g["out_blocksum_%d" % i])) < 1e-9
