"""GPU: the non-default analysis parameters of World.encode (frame_period, f0_floor / f0_ceil, channels_in_octave)
through the batched pipeline, against the oracle — nothing in the kernels may assume the 5 ms / 71-800 Hz defaults."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method,kw", [
    ("dio", dict(frame_period=10)),
    ("dio", dict(frame_period=2, f0_floor=90, f0_ceil=600)),
    ("dio", dict(frame_period=5, channels_in_octave=4, f0_floor=60, f0_ceil=500)),
    ("harvest", dict(frame_period=10)),
    ("harvest", dict(frame_period=2, f0_floor=90, f0_ceil=600)),
])
def test_parameters_match_oracle(method, kw):
    from oracle import api as oapi
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(70 + i, fs, 0.9 + 0.3 * i) for i in range(2)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method=method, **kw)
    dicts = enc.to_dicts()
    rng = np.random.RandomState(11)
    noise = [rng.randn(3 * len(x)) for x in xs]
    y, y_off = wb.decode_device(enc, noise=noise)
    y = y.cpu().numpy()
    for u, x in enumerate(xs):
        o = oapi.encode_np(fs, x, f0_method=method, **kw)
        d = dicts[u]
        assert np.array_equal(d["temporal_positions"], o["temporal_positions"])   # frame count and times: exact
        assert np.array_equal(d["vuv"], o["vuv"])
        assert rel_rms(d["f0"], o["f0"]) < 1e-8
        assert rel_rms(d["spectrogram"], o["spectrogram"]) < 1e-8
        assert rel_rms(d["aperiodicity"], o["aperiodicity"]) < 1e-8
        yo = oapi.decode_np(dict(d), noise=noise[u])["out"]
        seg = y[y_off[u]:y_off[u + 1]]
        assert len(seg) == len(yo)
        assert rel_rms(seg, yo) < 1e-8
    assert wb.rt.take_flags() == [0] * 16
