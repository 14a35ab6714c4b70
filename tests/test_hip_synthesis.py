"""GPU parity: wh_synthesis vs the golden fixture (seeded reference output) and the oracle."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _dat(g):
    return {"f0": g["d4c_f0_after"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy(),
            "spectrogram": g["ct_spectrogram"].copy(), "aperiodicity": g["d4c_aperiodicity"].copy(), "fs": int(g["fs"])}


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_synthesis_vs_golden(golden, tag):
    from world.synthesis import synthesis

    g = golden(tag)
    dat = _dat(g)
    np.random.seed(int(g["seed"]))
    y = synthesis(dat, dat)
    assert len(y) == len(g["syn_y"])  # bit-exact length (48 kHz float-arange quirk, Q9)
    # north_star tolerance 1e-4 relative RMS; same noise samples → ~1e-12
    assert rel_rms(y, g["syn_y"]) < 1e-9
    assert np.max(np.abs(y - g["syn_y"])) < 1e-10
    # the generator state after the call equals the reference's (same number of draws)
    after = np.random.randn()
    np.random.seed(int(g["seed"]))
    from oracle import resynth
    _, aux = resynth.synthesis_np(dat["f0"], dat["vuv"], dat["temporal_positions"], dat["spectrogram"],
                                  dat["aperiodicity"], dat["fs"], return_aux=True)
    assert after == np.random.randn()


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_synthesis_after_modifiers(golden, tag):
    """scale_pitch(1.5) + scale_duration(2.0) then decode (world/main.py:154-178)."""
    from world.synthesis import synthesis

    g = golden(tag)
    dat = _dat(g)
    dat["f0"] *= 1.5
    dat["temporal_positions"] *= 2.0
    np.random.seed(int(g["seed"]) + 1)
    y = synthesis(dat, dat)
    assert len(y) == int(g["mod_len"])
    assert np.max(np.abs(y[:2048] - g["mod_head"])) < 1e-10
    assert np.max(np.abs(y[-2048:] - g["mod_tail"])) < 1e-10
    assert np.max(np.abs(np.add.reduceat(y, np.arange(0, len(y), 256)) - g["mod_blocksum"])) < 1e-9


def test_synthesis_device_rng_statistics(golden):
    """Without host noise the Philox path is used: same deterministic (periodic) part, noise part of the
    same power.  Checked through the oracle fed with zero noise vs unit-variance noise."""
    from oracle import resynth
    from world import _hip
    from world.synthesis import synthesis_device, time_axis_params

    g = golden("syn16k")
    dat = _dat(g)
    rt = _hip.Runtime.get()
    tp = dat["temporal_positions"]
    ny, t0, dt = time_axis_params(tp, dat["fs"])
    batch = rt.make_batch([0, 0], [0, len(tp)])
    args = (rt.to_device(tp), rt.to_device(dat["f0"]), rt.to_device(dat["vuv"]),
            rt.to_device(np.ascontiguousarray(dat["spectrogram"].T)),
            rt.to_device(np.ascontiguousarray(dat["aperiodicity"].T)), dat["fs"], 1024, [ny], [t0], [dt])
    per = resynth.synthesis_np(dat["f0"], dat["vuv"], tp, dat["spectrogram"], dat["aperiodicity"], dat["fs"],
                               noise=np.zeros(4 * ny))  # deterministic part: oracle with an all-zero noise stream
    dev, ref, ys = [], [], []
    for seed in range(1, 7):  # a single realisation's noise power fluctuates by ~±25 %, so average
        y, _ = synthesis_device(rt, batch, *args, seed=seed)
        ys.append(y.cpu().numpy())
        dev.append(np.mean((ys[-1] - per) ** 2))
        np.random.seed(seed)
        r = resynth.synthesis_np(dat["f0"], dat["vuv"], tp, dat["spectrogram"], dat["aperiodicity"], dat["fs"])
        ref.append(np.mean((r - per) ** 2))
    assert not np.array_equal(ys[0], ys[1])
    assert 0.75 < np.mean(dev) / np.mean(ref) < 1.33
    assert rt.take_flags() == [0] * 16


def test_peak_normalisation_branches():
    """decode() divides by max|y| only where it exceeds 1 (world/main.py:209-212): a loud and a quiet utterance in
    one batch, against the same pair scaled on the host."""
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    import ctypes

    fs = 16000
    xs = [4.0 * synth_utterance(81, fs, 0.7), 0.2 * synth_utterance(82, fs, 0.6)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio")
    rng = np.random.RandomState(2)
    noise = [rng.randn(2 * len(x)) for x in xs]
    y, y_off = wb.decode_device(enc, noise=noise)
    y = y.cpu().numpy()
    loud, quiet = y[y_off[0]:y_off[1]], y[y_off[1]:y_off[2]]
    assert np.max(np.abs(loud)) == 1.0          # divided by its own maximum
    assert 0.0 < np.max(np.abs(quiet)) < 1.0    # left alone
    # the entry point itself on known data: [3, -6, 1.5] -> /6 ; [0.25, -0.5] untouched ; empty segment tolerated
    rt = wb.rt
    d = rt.to_device(np.array([3.0, -6.0, 1.5, 0.25, -0.5]))
    off = np.array([0, 3, 3, 5], dtype=np.int64)
    _hip.check(rt.lib.wh_peak_normalise(rt.ctx, rt.stream(), rt.ptr(d), off.ctypes.data_as(ctypes.c_void_p), 3))
    assert np.array_equal(d.cpu().numpy(), np.array([0.5, -1.0, 0.25, 0.25, -0.5]))
