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


def _philox_dump(wb, seed, ny_list):
    from world.synthesis import philox_normals

    # an utterance draws sum_i max(3, noise_size_i) <= ny + 3 * pulses samples; 2 * ny + 64 covers any f0 < fs / 2
    return [philox_normals(wb.rt, seed, u, 2 * ny + 64).cpu().numpy() for u, ny in enumerate(ny_list)]


@pytest.mark.parametrize("case", ["syn16k", "syn48k", "ragged3"])
def test_philox_decode_is_sample_exact(golden, case):
    """The decode bench.py times draws its noise on the device (Philox + Box-Muller).  wh_philox_normals exposes that
    stream; with it the seeded decode is checked sample by sample: (a) against the same kernels fed the dumped stream as
    host noise — the stream indexing of the two branches of response_kernel agrees, and (b) against the oracle's
    synthesis (world/synthesis.py:86-96 with np.random.randn replaced by the dumped stream)."""
    from oracle import api as oapi
    from world._synthetic import synth_utterance
    from world.batch import BatchEncoding, WorldBatch
    from world.synthesis import time_axis_params

    wb = WorldBatch()
    if case == "ragged3":
        fs = 16000
        xs = [synth_utterance(70, fs, 0.9), synth_utterance(71, fs, 0.37), synth_utterance(72, fs, 0.62)]
        enc = wb.encode(xs, fs, f0_method="dio")
        dats = enc.to_dicts()
    else:
        g = golden(case)
        dats = [dict(_dat(g), is_requiem=False)]
        enc = BatchEncoding.from_dicts(wb.rt, dats)
    fs = dats[0]["fs"]
    ny = [time_axis_params(d["temporal_positions"], fs)[0] for d in dats]
    for seed in (0, 12345):
        y_seed, off = wb.decode_device(enc, seed=seed)
        y_seed = y_seed.cpu().numpy()
        dump = _philox_dump(wb, seed, ny)
        for z in dump:  # a standard-normal stream, not zeros
            assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
        y_noise, off2 = wb.decode_device(enc, noise=dump)
        assert np.array_equal(off, off2)
        scale = np.max(np.abs(y_seed))
        # same arithmetic, overlap-add by FP64 atomics: equal up to the order of the adds
        assert np.max(np.abs(y_noise.cpu().numpy() - y_seed)) < 1e-13 * max(scale, 1.0)
        for u, d in enumerate(dats):
            yo = oapi.decode_np(dict(d), noise=dump[u])["out"]
            seg = y_seed[off[u]:off[u + 1]]
            assert len(seg) == len(yo)
            assert rel_rms(seg, yo) < 1e-9
            assert np.max(np.abs(seg - yo)) < 1e-9 * max(scale, 1.0)
    assert wb.rt.take_flags() == [0] * 16
    # another seed is another stream, another utterance index another stream of the same seed
    a, b = _philox_dump(wb, 1, [64, 64])
    assert not np.array_equal(a, b) and not np.array_equal(a, _philox_dump(wb, 2, [64])[0])


def test_short_host_noise_raises():
    """A host-supplied noise stream that does not cover an utterance's draws must raise WH_FLAG_NOISE_SHORT (the
    missing samples would otherwise read as zeros)."""
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_utterance(80, fs, 0.5), synth_utterance(81, fs, 0.5)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio")
    rng = np.random.RandomState(1)
    full = [rng.randn(2 * len(x) + 64) for x in xs]
    wb.decode_device(enc, noise=full)  # covers: no flag
    with pytest.raises(_hip.WorldHipError, match="noise"):
        wb.decode_device(enc, noise=[full[0], full[1][:1000]])
    assert wb.rt.take_flags() == [0] * 16  # read-and-cleared by the raise


def test_requiem_decode_with_device_seed_tables_is_sample_exact():
    """The batched Requiem decode's default seed tables are generated on the device (wh_requiem_seeds).  Dumped and
    handed to the oracle's synthesisRequiem (world/synthesisRequiem.py:12-141) they must reproduce the device decode
    sample by sample, the circular cursor chained across the utterances."""
    from oracle import resynth
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals_device

    fs = 16000
    xs = [synth_utterance(90, fs, 0.8), synth_utterance(91, fs, 0.45), synth_utterance(92, fs, 0.6)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio", is_requiem=True)
    seeds_d = get_seeds_signals_device(fs, seed=5)
    y, off = wb.decode_device(enc, seeds=seeds_d)
    y = y.cpu().numpy()
    seeds = {"pulse": seeds_d["pulse_d"].cpu().numpy(), "noise": seeds_d["noise_d"].cpu().numpy()}
    cursor = None
    for u, d in enumerate(enc.to_dicts()):
        yo, cursor = resynth.synthesis_requiem_np(d["f0"], d["vuv"], d["temporal_positions"], d["spectrogram"],
                                                  d["aperiodicity"], fs, seeds, cursor=cursor)
        peak = np.max(np.abs(yo))
        if peak > 1:
            yo = yo / peak
        seg = y[off[u]:off[u + 1]]
        assert len(seg) == len(yo)
        assert rel_rms(seg, yo) < 1e-9
    assert wb.rt.take_flags() == [0] * 16


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


def test_row_region_overflow_takes_the_pulse_capacity_retry():
    """The overlap-add rows of an utterance live in a region of 12 doubles per output sample (wh_synthesis.hip RunState);
    a pitch so high that the runs' rows do not fit — here a contour scaled to ~1.2-1.6 kHz at 16 kHz — raises
    WH_FLAG_PULSE_OVERFLOW from pulse_rows_kernel although the pulse slots themselves suffice, and decode_device's
    retry with the safe capacity (which sizes the region too) gives exactly what an explicit safe capacity gives."""
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.synthesis import safe_pulse_cap, time_axis_params

    fs = 16000
    xs = [synth_utterance(610, fs, 0.8), synth_utterance(611, fs, 0.6)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio")
    enc.scale_pitch(7.0)
    rng = np.random.RandomState(4)
    noise = [rng.randn(4 * len(x)) for x in xs]
    y, y_off = wb.decode_device(enc, noise=noise)                      # default capacity -> overflow -> retried inside
    fo = enc.batch.frame_off
    tp = enc.host_times()
    ny = [time_axis_params(tp[int(fo[u]):int(fo[u + 1])], fs)[0] for u in range(2)]
    y_safe, _ = wb.decode_device(enc, noise=noise, pulse_cap=safe_pulse_cap(ny))
    assert np.array_equal(y.cpu().numpy(), y_safe.cpu().numpy())
    assert np.all(np.isfinite(y.cpu().numpy())) and float(y.abs().max()) > 1e-3
    wb.decode_device(enc, noise=noise, check=False)                     # asynchronous: the condition is reported ...
    with pytest.raises(_hip.WorldHipError, match="pulse_cap"):
        wb.check()                                                      # ... by check()

