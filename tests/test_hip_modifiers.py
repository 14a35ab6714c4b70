"""GPU: the device-resident modifier pipeline — BatchEncoding.warp_spectrum / modify_duration
against reference output (tests/golden/golden_modifiers.npz, make_golden.py modifiers_fixture), the host facade's
versions of the same methods, and 16-bit WAV in / out around a resident batch."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _dat(g):
    return {"f0": g["d4c_f0_after"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy(),
            "spectrogram": g["ct_spectrogram"].copy(), "aperiodicity": g["d4c_aperiodicity"].copy(),
            "fs": int(g["fs"]), "is_requiem": False}


def test_warp_and_modify_duration_on_device_vs_reference(golden):
    from world import _hip
    from world.batch import BatchEncoding, WorldBatch
    from world.synthesis import synthesis_plan, time_axis_params

    g, m = golden("syn16k"), golden("modifiers")
    wb = WorldBatch()
    # the fixture utterance twice in one batch: both copies must come out like the reference's single run
    enc = BatchEncoding.from_dicts(wb.rt, [_dat(g), _dat(g)])
    enc.warp_spectrum(1.1)
    d = enc.to_dicts()
    for u in range(2):
        assert np.array_equal(d[u]["spectrogram"][:, ::30], m["warp_1p1_cols"])
        assert rel_rms(d[u]["spectrogram"].sum(axis=0), m["warp_1p1_colsum"]) < 1e-14
    enc.warp_spectrum(0.9)
    assert rel_rms(enc.to_dicts()[1]["spectrogram"].sum(axis=0), m["warp_then_0p9_colsum"]) < 1e-14
    enc.modify_duration(list(m["from_time"]), list(m["to_time"]))
    d = enc.to_dicts()
    for u in range(2):
        assert np.array_equal(d[u]["temporal_positions"], m["moddur_tp"])
    # decode with the reference's noise: as many randn draws as the reference makes, from the fixture's seed
    tp = d[0]["temporal_positions"]
    ny, t0, dt = time_axis_params(tp, enc.fs)
    assert ny == int(m["y_len"])
    one = BatchEncoding.from_dicts(wb.rt, [d[0]])
    _, draws = synthesis_plan(wb.rt, one.batch, one.temporal_positions, one.f0, one.vuv, one.fs, [ny], [t0], [dt],
                              ny // 2 + 16)
    np.random.seed(int(m["seed"]))
    noise = np.random.randn(int(draws[0]))
    y, y_off = wb.decode_device(enc, noise=[noise, noise])
    y = y.cpu().numpy()
    for u in range(2):
        seg = y[y_off[u]:y_off[u + 1]]
        assert len(seg) == int(m["y_len"])
        assert np.max(np.abs(seg[:4096] - m["y_head"])) < 1e-9
        assert np.max(np.abs(seg[-4096:] - m["y_tail"])) < 1e-9
        assert np.max(np.abs(np.add.reduceat(seg, np.arange(0, len(seg), 256)) - m["y_blocksum"])) < 1e-8
    assert wb.rt.take_flags() == [0] * 16
    assert isinstance(_hip.FLAG_MESSAGES, dict)


def test_host_facade_modifiers_vs_reference(golden):
    from world import main

    g, m = golden("syn16k"), golden("modifiers")
    W = main.World()
    dat = _dat(g)
    assert W.warp_spectrum(dat, 1.1) is dat
    assert np.array_equal(dat["spectrogram"][:, ::30], m["warp_1p1_cols"])
    W.warp_spectrum(dat, 0.9)
    assert W.modify_duration(dat, list(m["from_time"]), list(m["to_time"])) is None
    assert np.array_equal(dat["temporal_positions"], m["moddur_tp"])


def test_wav_in_wav_out(tmp_path):
    """int16 WAVs -> resident encode -> decode -> int16 WAVs: the PCM conversions follow the reference callers'
    conventions bit for bit, and the round trip equals the float path."""
    from scipy.io import wavfile

    from world import wavio
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    pcm = [np.round(synth_utterance(60 + i, fs, 0.5 + 0.2 * i) * 32767).astype(np.int16) for i in range(3)]
    paths = [tmp_path / ("in%d.wav" % i) for i in range(3)]
    for p, x in zip(paths, pcm):
        wavfile.write(str(p), fs, x)
    wb = WorldBatch()
    fs_r, enc = wavio.encode_wavs(paths, wb, f0_method="dio")
    assert fs_r == fs
    ref = wb.encode([x / (2 ** 15 - 1) for x in pcm], fs, f0_method="dio")  # example/prosody.py:13
    assert np.array_equal(enc.f0.cpu().numpy(), ref.f0.cpu().numpy())
    assert np.array_equal(enc.spectrogram.cpu().numpy(), ref.spectrogram.cpu().numpy())
    y, y_off = wb.decode_device(enc, seed=3)
    outs = [tmp_path / ("out%d.wav" % i) for i in range(3)]
    wavio.write_wavs(outs, fs, wb, y, y_off)
    yh = y.cpu().numpy()
    for u, p in enumerate(outs):
        fs_w, w = wavfile.read(str(p))
        assert fs_w == fs and w.dtype == np.int16
        assert np.array_equal(w, (yh[y_off[u]:y_off[u + 1]] * 2 ** 15).astype(np.int16))  # example/prosody.py:57
    # truncation toward zero on both sides of it, full-scale negative exact
    vals = np.array([-1.0, 0.99999, -0.5, 0.0, 3.1e-5, -3.1e-5, 0.25 + 1e-6, -0.25 - 1e-6])
    got = wb.to_pcm16(wb.rt.to_device(vals), [0, len(vals)])[0]
    assert np.array_equal(got, (vals * 2 ** 15).astype(np.int16))
    assert os.path.getsize(str(outs[0])) > 44
