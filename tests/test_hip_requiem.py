"""GPU parity: seeds (host), wh_synthesis_requiem and the World facade on BASELINE config 1."""
import os
import random

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_requiem_synthesis_vs_golden(golden, tag):
    from world import synthesisRequiem as sr
    from world.get_seeds_signals import get_seeds_signals

    g = golden(tag)
    fs = int(g["fs"])
    random.seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    seeds = get_seeds_signals(fs)
    assert np.max(np.abs(seeds["pulse"] - g["seeds_pulse"])) < 1e-15
    assert np.max(np.abs(seeds["noise"] - g["seeds_noise"])) < 1e-13
    f0r = np.where(g["dio_vuv"] == 0, 0.0, g["ct_f0_after"])
    dat = {"f0": f0r, "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy(),
           "spectrogram": g["ct_spectrogram"].copy(), "aperiodicity": g["req_band_ap"].copy(), "fs": fs}
    sr.generate_noise.current_index = None
    y = sr.synthesisRequiem(dat, dat, {"pulse": g["seeds_pulse"], "noise": g["seeds_noise"]})
    assert len(y) == len(g["req_y"])
    assert rel_rms(y, g["req_y"]) < 1e-9
    assert np.array_equal(np.asarray(sr.generate_noise.current_index), g["req_cursor"])
    # a second call continues from the stored cursor, like the reference
    from oracle import resynth
    y2 = sr.synthesisRequiem(dat, dat, {"pulse": g["seeds_pulse"], "noise": g["seeds_noise"]})
    yo, _ = resynth.synthesis_requiem_np(dat["f0"], dat["vuv"], dat["temporal_positions"], dat["spectrogram"],
                                         dat["aperiodicity"], fs, {"pulse": g["seeds_pulse"], "noise": g["seeds_noise"]},
                                         cursor=g["req_cursor"])
    assert rel_rms(y2, yo) < 1e-9


def test_world_facade_config1(golden):
    """BASELINE config 1 through the drop-in API: World().encode(f0_method='harvest') + decode on test-mwm.wav,
    both synthesis paths, seeded like the fixture."""
    from scipy.io import wavfile

    from world import main
    from world import synthesisRequiem as sr

    g = golden("mwm")
    fs, xi = wavfile.read(os.path.join(os.path.dirname(__file__), "golden", "test-mwm.wav"))
    x = xi / (2 ** 15 - 1)
    W = main.World()
    for req in (False, True):
        tag = "req" if req else "std"
        dat = W.encode(fs, x, f0_method="harvest", is_requiem=req)
        assert set(dat.keys()) == {"temporal_positions", "vuv", "fs", "f0", "aperiodicity", "ps spectrogram",
                                   "spectrogram", "is_requiem"}
        assert np.array_equal(dat["vuv"], g["vuv"])
        assert np.max(np.abs(dat["f0"] - g["f0"])) < 1e-6
        if not req:
            assert rel_rms(dat["spectrogram"].sum(axis=0), g["spec_colsum"]) < 1e-8
            assert rel_rms(dat["spectrogram"][:, g["cols"]], g["spec_cols"]) < 1e-8
            assert np.max(np.abs(dat["aperiodicity"][:, g["cols"]] - g["ap_cols"])) < 1e-7
        else:
            assert np.max(np.abs(dat["aperiodicity"] - g["req_band_ap"])) < 1e-6
        random.seed(int(g["seed"]))
        np.random.seed(int(g["seed"]))
        sr.generate_noise.current_index = None
        out = W.decode(dat)
        assert out is dat
        y = dat["out"]
        assert len(y) == int(g["out_len_" + tag])
        # north_star tolerance: 1e-4 relative RMS (chained encode→decode)
        assert np.max(np.abs(y[:4096] - g["out_head_" + tag])) < 1e-6
        assert np.max(np.abs(y[-4096:] - g["out_tail_" + tag])) < 1e-6
        assert np.max(np.abs(np.add.reduceat(y, np.arange(0, len(y), 256)) - g["out_blocksum_" + tag])) < 1e-5
