"""GPU: the drop-in facade over World.encode's arguments — the twelve cases of tests/_sweep_cases.py (search ranges, frame
periods of 1 / 2.5 / 3 / 10 / 12.5 ms, DIO's channels / target rate / allowed range, the fft_size override, 8 / 11.025 / 24 /
32 / 44.1 kHz, an int16-scaled waveform, a length on a filter-tile edge) — against the REFERENCE's outputs
(tests/golden/golden_sweep.npz: frame times and VUV exact, f0 1e-8, tensor sums, the seeded decode's block sums) and
against the oracle's full tensors (world/main.py:106-152,198-214)."""
import random

import numpy as np
import pytest

from _sweep_cases import sweep_cases, sweep_input
from conftest import rel_rms

pytestmark = pytest.mark.gpu
CASES = sweep_cases()


@pytest.mark.parametrize("i", range(len(CASES)))
def test_facade_case_against_reference_and_oracle(golden, i):
    from oracle import api as oapi
    from world import _hip
    from world import synthesisRequiem as sr
    from world._synthetic import synth_utterance
    from world.main import World

    g = golden("sweep")
    _, fs, _, _, kw = CASES[i]
    x = sweep_input(synth_utterance, CASES[i])
    W = World()
    dat = W.encode(fs, x.copy(), **kw)
    # the reference's own numbers
    assert np.array_equal(dat["temporal_positions"], g["tp_%d" % i])
    assert np.array_equal(dat["vuv"], g["vuv_%d" % i])
    assert np.allclose(dat["f0"], g["f0_%d" % i], rtol=1e-8, atol=0)
    assert list(dat["spectrogram"].shape) == list(g["spec_shape_%d" % i])
    assert rel_rms(dat["spectrogram"].sum(axis=0), g["spec_colsum_%d" % i]) < 1e-8
    assert rel_rms(dat["spectrogram"].sum(axis=1), g["spec_rowsum_%d" % i]) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=0) - g["ap_colsum_%d" % i])) < 1e-5
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=1) - g["ap_rowsum_%d" % i])) < 1e-5
    # the oracle's full tensors
    o = oapi.encode_np(fs, x.copy(), **kw)
    assert rel_rms(dat["spectrogram"], o["spectrogram"]) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"] - o["aperiodicity"])) < (1e-6 if kw.get("is_requiem") else 1e-7)
    # the seeded decode (np.random / random drawn in the reference's order by the mirror)
    random.seed(int(g["seed"]) + 200 + i)
    np.random.seed(int(g["seed"]) + 200 + i)
    sr.generate_noise.current_index = None
    y = W.decode(dict(dat))["out"]
    assert len(y) == int(g["out_len_%d" % i])
    bs = np.add.reduceat(y, np.arange(0, len(y), 256))
    scale = max(1.0, float(np.max(np.abs(g["out_blocksum_%d" % i]))))
    assert np.max(np.abs(bs - g["out_blocksum_%d" % i])) < 1e-7 * scale
    assert _hip.Runtime.get().take_flags() == [0] * 16
