"""GPU parity against the REFERENCE's own output on the 12 differential draws (tests/golden/golden_differential.npz,
`make_golden.py differential`): World().encode + seeded World().decode through the drop-in facade at 8 / 16 / 22.05 /
44.1 / 48 kHz with dio / harvest / swipe, D4C / Requiem, frame periods 4 and 5 ms, f0 floors 71 and 90 Hz.
Tolerances: vuv, frame and sample counts exact; f0 1e-9 relative (SWIPE': at most two frames one 1/768-octave step
off, see test_hip_swipe.py); tensor sums 1e-8; waveform block sums 1e-6 (north_star: 1e-4 relative RMS)."""
import random

import numpy as np
import pytest

from conftest import rel_rms
from test_oracle_differential import N_DRAWS, draw_args

pytestmark = pytest.mark.gpu


def _encode_checks(g, i, dat, method):
    assert np.array_equal(dat["vuv"], g["vuv_%d" % i])
    ref_f0 = g["f0_%d" % i]
    rel = np.abs(dat["f0"] - ref_f0) / np.maximum(ref_f0, 1.0)
    same = rel < 1e-9
    if method == "swipe":
        assert (~same).sum() <= 2 and np.all(rel < 1e-3), float(rel.max())
    else:
        assert np.all(same), float(rel.max())
    # spectra depend on f0: compare the frames whose f0 agrees
    assert rel_rms(dat["spectrogram"].sum(axis=0)[same], g["spec_colsum_%d" % i][same]) < 1e-8
    k_bins = dat["aperiodicity"].shape[0]
    # column sums over K rows: mean per-row error below 1e-8 for D4C's amplitude rows (it sums ~1000 FFT bins per band
    # in another order), below 1e-7 for Requiem's band rows, which are in dB (-60 ... 0; test_hip_d4c.py uses 1e-6)
    per_row = 1e-7 if dat["is_requiem"] else 1e-8
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=0)[same] - g["ap_colsum_%d" % i][same])) < per_row * k_bins
    return bool(np.all(same))


@pytest.mark.parametrize("i", range(N_DRAWS))
def test_facade_vs_reference_draw(golden, i):
    from world import main
    from world import synthesisRequiem as sr
    from world._synthetic import synth_utterance

    g = golden("differential")
    u, fs, sec, kw = draw_args(g, i)
    x = synth_utterance(u, fs, sec)
    W = main.World()
    dat = W.encode(fs, x.copy(), **kw)
    all_same = _encode_checks(g, i, dat, kw["f0_method"])
    random.seed(int(g["seed"]) + i)
    np.random.seed(int(g["seed"]) + i)
    sr.generate_noise.current_index = None
    y = W.decode(dat)["out"]
    assert len(y) == int(g["out_len_%d" % i])
    if all_same:
        bs = np.add.reduceat(y, np.arange(0, len(y), 256))
        assert np.max(np.abs(bs - g["out_blocksum_%d" % i])) < 1e-6


@pytest.mark.parametrize("i", [1, 3, 6, 10])
def test_seeded_chain_with_reference_rng_consumption(golden, i):
    """One seed before encode(), none before decode(): the reference's CheapTrick draws rand(K) per frame from the
    global stream, so synthesis sees a shifted generator.  With world.cheaptrick.CONSUME_REFERENCE_RNG the drop-in
    walks the stream the same way and the seeded encode -> decode chain matches the reference's sample sums."""
    from world import cheaptrick as ct
    from world import main
    from world import synthesisRequiem as sr
    from world._synthetic import synth_utterance

    g = golden("differential")
    u, fs, sec, kw = draw_args(g, i)
    assert kw["f0_method"] != "swipe"
    x = synth_utterance(u, fs, sec)
    W = main.World()
    ct.CONSUME_REFERENCE_RNG = True
    try:
        random.seed(int(g["seed"]) + 100 + i)
        np.random.seed(int(g["seed"]) + 100 + i)
        sr.generate_noise.current_index = None
        y = W.decode(W.encode(fs, x.copy(), **kw))["out"]
    finally:
        ct.CONSUME_REFERENCE_RNG = False
    bs = np.add.reduceat(y, np.arange(0, len(y), 256))
    assert len(bs) == len(g["chain_blocksum_%d" % i])
    assert np.max(np.abs(bs - g["chain_blocksum_%d" % i])) < 1e-6
