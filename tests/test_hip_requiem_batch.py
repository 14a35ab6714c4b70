"""GPU parity of the BATCHED Requiem decode (the entry point bench.py --config 4 times):
WorldBatch.decode_device on an is_requiem encoding == consecutive reference-style synthesisRequiem calls that
share the persistent noise cursor (world/synthesisRequiem.py:131-141, world/main.py:205-206)."""
import random

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _oracle_chain(dicts, fs, seeds, cursor=None):
    from oracle import resynth

    outs = []
    for d in dicts:
        y, cursor = resynth.synthesis_requiem_np(d["f0"], d["vuv"], d["temporal_positions"], d["spectrogram"],
                                                 d["aperiodicity"], fs, seeds, cursor)
        m = np.max(np.abs(y))
        outs.append(y / m if m > 1.0 else y)
    return outs, cursor


@pytest.mark.parametrize("method", ["dio", "harvest"])
def test_batched_requiem_decode_matches_chained_oracle(method):
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals

    fs = 16000
    xs = [synth_utterance(30 + i, fs, s) for i, s in enumerate((0.9, 0.55, 1.3, 0.7))]  # ragged
    random.seed(7)
    np.random.seed(7)
    seeds = get_seeds_signals(fs)
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method=method, is_requiem=True)
    y, y_off = wb.decode_device(enc, seeds=seeds)
    y = y.cpu().numpy()
    dicts = enc.to_dicts()
    ref, cur = _oracle_chain(dicts, fs, seeds)
    for u in range(len(xs)):
        seg = y[y_off[u]:y_off[u + 1]]
        assert len(seg) == len(ref[u])
        assert rel_rms(seg, ref[u]) < 1e-8, u
    assert np.array_equal(np.asarray(enc.requiem_cursor, dtype=np.float64), np.asarray(cur, dtype=np.float64))
    # a second batch continues where the first stopped, like a second round of reference calls would
    y2, y2_off = wb.decode_device(enc, seeds=seeds, cursor=cur)
    ref2, _ = _oracle_chain(dicts, fs, seeds, cursor=np.array(cur, dtype=np.float64))
    y2 = y2.cpu().numpy()
    for u in range(len(xs)):
        assert rel_rms(y2[y2_off[u]:y2_off[u + 1]], ref2[u]) < 1e-8, u
    assert wb.rt.take_flags() == [0] * 16


def test_requiem_decode_with_a_noise_table_that_is_not_a_power_of_two():
    """The circular read of the band noises (synthesisRequiem.py:131-141) takes the index modulo the table length; the
    default lengths are powers of two (a mask on the device), any other length takes the general path.  Short
    utterances with a low pitch also put pulses within half a seed length of both ends, where the reference's clipped
    fancy-index assignment keeps only the last tap written (the gathered excitation reproduces that per sample)."""
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals

    fs = 16000
    xs = [synth_utterance(70 + i, fs, s) for i, s in enumerate((0.31, 0.52, 0.2))]
    random.seed(11)
    np.random.seed(11)
    seeds = get_seeds_signals(fs, noise_length=6000)
    assert seeds['noise'].shape[0] == 6000
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio", is_requiem=True)
    start = np.tile(np.array([5990.0, 17.0, 3000.0]), (len(xs), 1))[:, :seeds['noise'].shape[1]]
    y, y_off = wb.decode_device(enc, seeds=seeds, cursor=start[0])
    y = y.cpu().numpy()
    ref, _ = _oracle_chain(enc.to_dicts(), fs, seeds, cursor=start[0].copy())
    for u in range(len(xs)):
        assert rel_rms(y[y_off[u]:y_off[u + 1]], ref[u]) < 1e-8, u
    assert wb.rt.take_flags() == [0] * 16


def test_batched_requiem_decode_after_modifiers_and_vs_single():
    """scale_pitch / scale_duration on the resident encoding, then the batched decode: per-utterance hops and
    output lengths follow the host formulas (Q9, Q11) and each utterance equals the single-utterance drop-in."""
    from world import synthesisRequiem as sr
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals
    from world.synthesis import time_axis_params

    fs = 16000
    xs = [synth_utterance(44 + i, fs, 0.6 + 0.3 * i) for i in range(3)]
    random.seed(3)
    np.random.seed(3)
    seeds = get_seeds_signals(fs)
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio", is_requiem=True)
    enc.scale_pitch(1.3).scale_duration(1.7)
    y, y_off = wb.decode_device(enc, seeds=seeds)
    y = y.cpu().numpy()
    dicts = enc.to_dicts()
    sr.generate_noise.current_index = None
    for u, d in enumerate(dicts):
        assert y_off[u + 1] - y_off[u] == time_axis_params(d["temporal_positions"], fs)[0]
        single = sr.synthesisRequiem(d, d, seeds)  # persistent cursor carries over, like the batch
        m = np.max(np.abs(single))
        if m > 1.0:
            single = single / m
        assert rel_rms(y[y_off[u]:y_off[u + 1]], single) < 1e-10, u
    ref, _ = _oracle_chain(dicts, fs, seeds)
    for u in range(len(xs)):
        assert rel_rms(y[y_off[u]:y_off[u + 1]], ref[u]) < 1e-8, u


def test_pulse_capacity_overflow_is_not_silent():
    """a mean pulse rate above fs/8 overflows the default pulse capacity.  The checked decode re-runs
    with the safe bound and matches the oracle; an explicit too-small pulse_cap raises instead of truncating."""
    from oracle import api as oapi
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 8000
    x = synth_utterance(61, fs, 1.0)
    wb = WorldBatch()
    enc = wb.encode([x], fs, f0_method="dio")
    enc.f0.fill_(1500.0)  # 1500 pulses/s on every frame: above the default capacity of fs/8 = 1000 pulses/s
    enc.vuv.fill_(1.0)
    rng = np.random.RandomState(2)
    noise = [rng.randn(6 * len(x))]
    y, y_off = wb.decode_device(enc, noise=noise)
    d = enc.to_dicts()[0]
    yo = oapi.decode_np(dict(d), noise=noise[0])["out"]
    assert len(yo) == y_off[1]
    assert rel_rms(y.cpu().numpy(), yo) < 1e-8
    with pytest.raises(_hip.WorldHipError):
        wb.decode_device(enc, noise=noise, pulse_cap=64)
    assert wb.rt.take_flags() == [0] * 16
