"""Oracle faithfulness beyond the stage fixtures (CPU): tests/golden/golden_differential.npz is written by
`make_golden.py differential`, which runs the UNMODIFIED reference's World.encode + seeded decode next to
oracle/api.py on 12 seeded draws of (utterance, fs in 8-48 kHz, dio / harvest / swipe, D4C / Requiem, frame period,
f0 floor) that no other fixture touches.  It stores the worst reference-vs-oracle error per tensor as measured there,
and the reference's own f0 / vuv / tensor sums per draw — re-checked here against the oracle without the reference."""
import random

import numpy as np
import pytest

from conftest import rel_rms
from oracle import api

N_DRAWS = 12


def draw_args(g, i):
    kw = dict(f0_method=str(g["draw_method"][i]), is_requiem=bool(g["draw_requiem"][i]),
              frame_period=int(g["draw_frame_period"][i]), f0_floor=int(g["draw_f0_floor"][i]))
    return int(g["draw_utt"][i]), int(g["draw_fs"][i]), float(g["draw_seconds"][i]), kw


def test_recorded_worst_errors(golden):
    g = golden("differential")
    assert len(g["draw_utt"]) == N_DRAWS
    assert set(g["draw_method"]) == {"dio", "harvest", "swipe"} and set(g["draw_fs"]) == {8000, 16000, 22050, 44100, 48000}
    assert g["draw_requiem"].sum() >= 4
    # discrete decisions: identical on every draw
    for k in ("vuv_mismatch", "frames_mismatch", "out_len_mismatch", "tp_maxabs"):
        assert float(g["worst_" + k]) == 0.0, k
    assert float(g["worst_f0_maxrel"]) < 1e-12
    assert float(g["worst_spectrogram_relrms"]) < 1e-12   # the reference's rand*eps dither vs eps/2
    assert float(g["worst_aperiodicity_maxabs"]) < 1e-10
    assert float(g["worst_out_relrms"]) < 1e-10


@pytest.mark.parametrize("i", range(N_DRAWS))
def test_oracle_reproduces_reference_draw(golden, i):
    from world._synthetic import synth_utterance

    g = golden("differential")
    u, fs, sec, kw = draw_args(g, i)
    x = synth_utterance(u, fs, sec)
    dat = api.encode_np(fs, x, **kw)
    assert np.array_equal(dat["vuv"], g["vuv_%d" % i])
    assert np.allclose(dat["f0"], g["f0_%d" % i], rtol=1e-12, atol=0)
    assert rel_rms(dat["spectrogram"].sum(axis=0), g["spec_colsum_%d" % i]) < 1e-12
    assert rel_rms(dat["spectrogram"].sum(axis=1), g["spec_rowsum_%d" % i]) < 1e-12
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=0) - g["ap_colsum_%d" % i])) < 1e-8
    assert np.max(np.abs(dat["aperiodicity"].sum(axis=1) - g["ap_rowsum_%d" % i])) < 1e-8
    random.seed(int(g["seed"]) + i)
    np.random.seed(int(g["seed"]) + i)
    y = api.decode_np(dat)["out"]
    assert len(y) == int(g["out_len_%d" % i])
    assert np.max(np.abs(np.add.reduceat(y, np.arange(0, len(y), 256)) - g["out_blocksum_%d" % i])) < 1e-9
