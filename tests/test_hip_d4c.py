"""GPU parity: wh_d4c / wh_d4c_requiem vs the golden fixture (reference output) and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_d4c(golden, tag):
    from world.d4c import d4c

    g = golden(tag)
    fs = int(g["fs"])
    src = {"f0": g["ct_f0_after"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy()}
    out = d4c(g["x"], fs, src)
    assert out is src  # same dict object, keys added (Q6)
    assert np.array_equal(src["f0"], g["d4c_f0_after"])
    # love-train decisions are discrete: the set of gated frames must match exactly
    assert np.array_equal(out["coarse_ap"] != 0, g["d4c_coarse"] != 0)
    assert np.array_equal((out["aperiodicity"] < 0.999999).any(axis=0), (g["d4c_aperiodicity"] < 0.999999).any(axis=0))
    assert np.max(np.abs(out["coarse_ap"] - g["d4c_coarse"])) < 1e-6   # dB
    assert np.max(np.abs(out["aperiodicity"] - g["d4c_aperiodicity"])) < 1e-7


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_d4c_requiem(golden, tag):
    from world.d4cRequiem import d4cRequiem

    g = golden(tag)
    fs = int(g["fs"])
    src = {"f0": g["ct_f0_after"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy()}
    out = d4cRequiem(g["x"], fs, src)
    assert out["aperiodicity"].shape == g["req_band_ap"].shape
    assert np.max(np.abs(out["aperiodicity"] - g["req_band_ap"])) < 1e-6  # dB


@pytest.mark.parametrize("fs", [16000, 48000])
def test_d4c_rank_select_wide_dynamic_range(fs):
    """The band stage sums the smallest N/2 - boundary of the K power bins (world/d4c.py:206-208); the kernel selects
    by IEEE exponent in rounds of 8 octaves.  Inputs whose band spectra spread over far more than 8 octaves — a clean
    click train, a smoothed one, and near-silence with one loud burst — must walk the further rounds and still agree with
    the oracle (which sorts).  These signals are also ill-conditioned for D4C as such (exact zeros between clicks: the
    centroid is divided by a smoothed power that is 1e-10 of its peak), so kernel and oracle agree to ~1e-5 ... 1e-4 dB
    here rather than to the 1e-9 of speech-like input — measured identically with the round-2 kernel, i.e. not a
    property of the selection; a wrong selection (an exponent bin dropped or kept too many) moves the result by 0.1 dB
    and more.  Tolerance: 1e-3 dB."""
    from oracle import aperiodicity as oap
    from oracle import common as C
    from world.d4c import d4c
    from world.d4cRequiem import d4cRequiem

    n = int(0.5 * fs)
    t = np.arange(n) / fs
    rng = np.random.RandomState(5)
    # (checked on the oracle's spectra: on the frames that pass the love-train gate the 22 largest bins of these inputs
    # span up to 12 / 14 / 20 octaves — two and three selection rounds)
    clicks = np.zeros(n)
    clicks[:: int(fs / 110)] = 0.8
    soft = np.convolve(clicks, np.hanning(9), mode="same") + 1e-6 * rng.randn(n)
    clicks = clicks + 1e-5 * rng.randn(n)
    burst = 1e-6 * rng.randn(n)
    nb = int(0.025 * fs)
    burst[n // 2:n // 2 + nb] += 0.5 * np.sin(2 * np.pi * 200.0 * t[:nb])
    for x in (clicks, soft, burst):
        nf = C.frame_count(len(x), fs, 5)
        tp = C.frame_times(nf, 5)
        f0 = np.full(nf, 115.0)
        vuv = np.ones(nf)
        want_ap, want_coarse, _ = oap.d4c_np(x, fs, f0.copy(), vuv, tp)
        assert np.isfinite(want_coarse).all() and (want_coarse != 0).any()
        src = {"f0": f0.copy(), "vuv": vuv.copy(), "temporal_positions": tp.copy()}
        got = d4c(x, fs, src)
        assert np.array_equal(got["coarse_ap"] != 0, want_coarse != 0)  # the same frames pass the gate
        assert np.max(np.abs(got["coarse_ap"] - want_coarse)) < 1e-3
        assert np.max(np.abs(got["aperiodicity"] - want_ap)) < 1e-4
        want_band, _ = oap.d4c_requiem_np(x, fs, f0.copy(), vuv, tp)
        src = {"f0": f0.copy(), "vuv": vuv.copy(), "temporal_positions": tp.copy()}
        got_band = d4cRequiem(x, fs, src)["aperiodicity"]
        assert np.isfinite(want_band).all()
        assert np.max(np.abs(got_band - want_band)) < 1e-3


def test_d4c_96k_needs_8192_point_transforms(golden):
    """Beyond ~54 kHz D4C's transform (d4c.py:20) and the love-train gate's (d4c.py:75) are 8192 points: the 96 kHz
    fixture (reference: harvest -> cheaptrick -> d4c / d4cRequiem)."""
    from world.d4c import d4c
    from world.d4cRequiem import d4cRequiem

    g = golden("syn96k")
    fs = int(g["fs"])
    src = {"f0": g["ct_f0_after"].copy(), "vuv": g["harvest_vuv"].copy(), "temporal_positions": g["tp"].copy()}
    out = d4c(g["x"], fs, src)
    assert np.array_equal(src["f0"], g["d4c_f0_after"])
    assert np.array_equal(out["coarse_ap"] != 0, g["d4c_coarse"] != 0)
    assert np.max(np.abs(out["coarse_ap"] - g["d4c_coarse"])) < 1e-6
    assert np.max(np.abs(out["aperiodicity"] - g["d4c_aperiodicity"])) < 1e-7
    src = {"f0": g["ct_f0_after"].copy(), "vuv": g["harvest_vuv"].copy(), "temporal_positions": g["tp"].copy()}
    rq = d4cRequiem(g["x"], fs, src)
    assert rq["aperiodicity"].shape == g["req_band_ap"].shape
    assert np.max(np.abs(rq["aperiodicity"] - g["req_band_ap"])) < 1e-6


@pytest.mark.parametrize("fs,requiem", [(16000, False), (16000, True), (48000, False)])
def test_love_train_gate_around_its_threshold(fs, requiem):
    """the voicing gate s1 / s2 > 0.85 (d4c.py:68-88) is a discrete decision read off the kernel's transform.
    A harmonic source below 4 kHz plus a band of noise between 4 and 7.9 kHz whose gain sweeps through the utterance
    walks the ratio across 0.85 in small steps: every frame whose ratio (oracle) lies further than 1e-9 from the
    threshold must be gated exactly as the oracle gates it — in the fused kernel (d4c, transform sizes equal), in
    love_train_kernel (D4C-Requiem at 16 kHz: 2048 against 1024) and at 48 kHz — whatever the translation unit's FMA
    contraction; the sweep must really pass the threshold, with frames within 1e-2 of it on both sides."""
    from oracle import aperiodicity as oap
    from world.d4c import d4c
    from world.d4cRequiem import d4cRequiem

    rng = np.random.RandomState(12)
    n = int(2.0 * fs)
    t = np.arange(n) / fs
    f0 = 140.0
    low = sum(np.sin(2 * np.pi * f0 * h * t + 0.3 * h) / h for h in range(1, int(3500 / f0)))
    spec = np.fft.rfft(rng.randn(n))
    fr = np.fft.rfftfreq(n, 1 / fs)
    spec[(fr < 4300) | (fr > 7600)] = 0.0
    high = np.fft.irfft(spec, n)
    high *= np.sqrt(np.mean(low ** 2) / np.mean(high ** 2))
    gain = np.linspace(0.15, 0.75, n)  # power ratio low / (low + g^2 high): ~0.98 down to ~0.64
    x = 0.2 * (low + gain * high)
    tp = np.arange(int(1000 * n / fs / 5 + 1)) * 0.005
    f0v = np.full(len(tp), f0)
    vuv = np.ones(len(tp))
    # the oracle's ratio per frame (love_train's own arithmetic, world/d4c.py:68-88)
    nfft = oap._pow2_at_least(3 * fs / 40.0 + 1)
    b0, b1, b2 = (int(np.ceil(v / (fs / nfft)) + 1) for v in (100, 4000, 7900))
    wave, _, _ = oap._window_frames(x, fs, np.maximum(f0v, 40.0), tp, 1.5, True)
    p = np.abs(np.fft.fft(wave, nfft, axis=1)) ** 2
    p[:, :b0] = 0.0
    cum = np.cumsum(p, axis=1)
    ratio = cum[:, b1 - 1] / cum[:, b2 - 1]
    want = ratio > 0.85
    assert want.any() and (~want).any()
    assert ((ratio > 0.85) & (ratio < 0.86)).any() and ((ratio < 0.85) & (ratio > 0.84)).any()
    src = {"f0": f0v.copy(), "vuv": vuv.copy(), "temporal_positions": tp.copy()}
    if requiem:
        out = d4cRequiem(x, fs, src)["aperiodicity"]
        got = (out[1:-1] < -1e-9).any(axis=0)  # an ungated frame keeps 0 dB... in every band
        ref, _ = oap.d4c_requiem_np(x, fs, f0v.copy(), vuv, tp)
        want_out = (ref[1:-1] < -1e-9).any(axis=0)
    else:
        out = d4c(x, fs, src)
        got = (out["coarse_ap"] != 0).any(axis=0)
        ref_ap, ref_coarse, _ = oap.d4c_np(x, fs, f0v.copy(), vuv, tp)
        want_out = (ref_coarse != 0).any(axis=0)
    clear = np.abs(ratio - 0.85) > 1e-9
    assert clear.sum() >= len(ratio) - 2
    assert np.array_equal(got[clear], want_out[clear])
    # (the gate is the only way a frame of this signal ends up without aperiodicity)
    assert np.array_equal(want_out[clear], want[clear])
