"""GPU: Requiem seed signals generated on the device (wh_requiem_seeds) — SURVEY.md 8(f)-4.

What is exact: the band pulses (deterministic) against the host get_seeds_signals(); the noise seeds against NumPy's
FFT product evaluated on the SAME velvet noise (the kernel sums the circular convolution directly).
What is statistical (the velvet noise draws from Philox, not from Python's `random` / NumPy's global stream): its
construction invariants (one +-2 impulse per 4-sample cell, signs balanced inside every segment, segment lengths
from the reference's three short periods), band energies of the noise seeds against host-generated ones, and the level
of a Requiem decode that uses them."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fs", [16000, 48000])
def test_device_seeds_exact_parts(fs):
    from world.get_seeds_signals import get_seeds_signals, get_seeds_signals_device

    random.seed(1)
    np.random.seed(1)
    host = get_seeds_signals(fs)
    dev = get_seeds_signals_device(fs, seed=5, want_velvet=True)
    pulse = dev["pulse_d"].cpu().numpy()
    noise = dev["noise_d"].cpu().numpy()
    velvet = dev["velvet_d"].cpu().numpy()
    assert pulse.shape == host["pulse"].shape and noise.shape == host["noise"].shape
    assert np.max(np.abs(pulse - host["pulse"])) < 1e-14
    # noise seeds == the reference's FFT product on this velvet noise; band 0 from the pulse BEFORE its DC correction
    n = len(velvet)
    spec_n = np.fft.fft(velvet, n)
    for b in range(pulse.shape[1]):
        p = pulse[:, b]
        if b == 0:
            from scipy.signal.windows import hann
            h = hann(len(p) + 2)[1:-1]
            # undo pulse[:,0] -= mean(raw) * h / mean(h): mean(corrected) = 0, so recover raw from the fixture-free identity
            raw = host["pulse"][:, 0] + 0  # corrected host pulse == corrected device pulse (checked above)
            # raw = corrected + c*h with c chosen so that the reference's formula maps raw -> corrected; any c works for
            # the check below only if the device used the same raw, so rebuild it exactly like the reference does
            w = np.arange(len(p) // 2 + 1) * fs / len(p)
            shape = 0.5 + 0.5 * np.cos(((w - 0) / 6000) * 2 * np.pi)
            shape[w > 3000] = 0
            shape[w < -3000] = 0
            p = np.fft.fftshift(np.fft.ifft(np.r_[shape, shape[-2:0:-1]]).real)
            assert np.max(np.abs((p - np.mean(p) * h / np.mean(h)) - raw)) < 1e-14
        ref = np.fft.ifft(spec_n * np.fft.fft(p, n)).real
        assert np.max(np.abs(noise[:, b] - ref)) < 1e-11


@pytest.mark.parametrize("fs", [16000, 22050, 48000])
def test_device_velvet_noise_construction_and_statistics(fs):
    from world.get_seeds_signals import get_seeds_signals, get_seeds_signals_device

    dev = get_seeds_signals_device(fs, seed=11, want_velvet=True)
    velvet = dev["velvet_d"].cpu().numpy()
    n = len(velvet)
    assert set(np.unique(velvet)) <= {-2.0, 0.0, 2.0}
    lens = [int(8 * (p * fs / 48000 + 0.5)) for p in (8, 30, 60)]
    # the noise must be a concatenation of segments of the three lengths, each with exactly one impulse in every
    # 4-sample cell, zeros behind the last cell and balanced signs (cells // 2 positive).  A short segment can look
    # like the head of a long one, so feasibility is decided backwards over all split points (dynamic programme).
    def valid(at, ln):
        cells = ln // 4
        seg = velvet[at:at + ln]
        if len(seg) < ln:  # the last segment is cut off at n
            seg = np.r_[seg, np.zeros(ln - len(seg))]
            body = seg[:4 * cells].reshape(cells, 4)
            return bool(np.all(seg[4 * cells:] == 0) and np.all((body != 0).sum(axis=1) <= 1))
        body = seg[:4 * cells].reshape(cells, 4)
        return bool(np.all(seg[4 * cells:] == 0) and np.all((body != 0).sum(axis=1) == 1)
                    and (body > 0).sum() == cells // 2)

    feasible = np.zeros(n + max(lens) + 1, dtype=bool)
    feasible[n - 1:] = True  # the reference stops once index >= N - 1
    used = set()
    for at in range(n - 2, -1, -1):
        for ln in lens:
            if feasible[at + ln] and valid(at, ln):
                feasible[at] = True
                used.add(ln)
    assert feasible[0]
    assert len(used) >= 2  # the segment lengths vary
    nz = np.count_nonzero(velvet)
    assert 0.23 * n < nz <= 0.25 * n + 1
    # statistics against host-generated seeds (different RNG, same construction): band RMS within 10 %
    noise = dev["noise_d"].cpu().numpy()
    rms = []
    for s in range(4):
        random.seed(100 + s)
        np.random.seed(100 + s)
        rms.append(np.sqrt(np.mean(get_seeds_signals(fs)["noise"] ** 2, axis=0)))
    rms = np.mean(rms, axis=0)
    dev_rms = np.sqrt(np.mean(noise ** 2, axis=0))
    assert np.all(np.abs(dev_rms / rms - 1) < 0.10), (dev_rms, rms)
    assert np.all(np.abs(np.mean(noise, axis=0)) < 0.05 * dev_rms + 1e-3)
    # two seeds give different noise, one seed gives the same noise
    again = get_seeds_signals_device(fs, seed=11)["noise_d"].cpu().numpy()
    other = get_seeds_signals_device(fs, seed=12)["noise_d"].cpu().numpy()
    assert np.array_equal(again, noise) and not np.array_equal(other, noise)


def test_requiem_decode_with_device_seeds():
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals, get_seeds_signals_device

    fs = 16000
    xs = [synth_utterance(33 + i, fs, 0.8) for i in range(2)]
    wb = WorldBatch()
    enc = wb.encode(xs, fs, f0_method="dio", is_requiem=True)
    random.seed(2)
    np.random.seed(2)
    y_host, off = wb.decode_device(enc, seeds=get_seeds_signals(fs))
    y_dev, off2 = wb.decode_device(enc, seeds=get_seeds_signals_device(fs, seed=3))
    assert np.array_equal(off, off2)
    a, b = y_host.cpu().numpy(), y_dev.cpu().numpy()
    assert np.all(np.isfinite(b))
    for u in range(2):
        ra = np.sqrt(np.mean(a[off[u]:off[u + 1]] ** 2))
        rb = np.sqrt(np.mean(b[off[u]:off[u + 1]] ** 2))
        assert abs(rb / ra - 1) < 0.1  # the voiced part is deterministic, the noise part has the same level
    assert wb.rt.take_flags() == [0] * 16


@pytest.mark.parametrize("fs", [16000, 48000])
def test_device_velvet_noise_whiteness(fs):
    """The velvet noise must be as white as the reference's: its power spectrum flat (spectral flatness = geometric /
    arithmetic mean of the periodogram averaged over 64 blocks) and its autocorrelation free of structure beyond what
    the construction itself puts there (one impulse per 4-sample cell: lags 1-3 are slightly negative).  Compared with
    host-generated velvet noise of the same construction, lag by lag."""
    from world.get_seeds_signals import get_seeds_signals_device

    from world.get_seeds_signals import _modified_velvet_noise

    def host_velvet(n, seed):  # the host mirror of the reference's construction, on the reference's RNG streams
        random.seed(seed)
        np.random.seed(seed)
        return _modified_velvet_noise(n, fs)

    def stats(v):
        blocks = v[:len(v) // 64 * 64].reshape(64, -1)
        psd = np.mean(np.abs(np.fft.rfft(blocks, axis=1)) ** 2, axis=0)[1:]
        flat = np.exp(np.mean(np.log(psd))) / np.mean(psd)
        v0 = v - v.mean()
        ac = np.array([np.dot(v0[:-k], v0[k:]) for k in range(1, 33)]) / np.dot(v0, v0)
        return flat, ac

    dev = get_seeds_signals_device(fs, seed=21, want_velvet=True)["velvet_d"].cpu().numpy()
    flat_d, ac_d = stats(dev)
    refs = [stats(host_velvet(len(dev), 300 + s)) for s in range(6)]
    flat_h = np.mean([r[0] for r in refs])
    ac_h = np.mean([r[1] for r in refs], axis=0)
    ac_sd = np.std([r[1] for r in refs], axis=0) + 1.0 / np.sqrt(len(dev) / 4)
    assert flat_d > 0.97 * flat_h and flat_d > 0.9, (flat_d, flat_h)
    assert np.all(np.abs(ac_d - ac_h) < 5 * ac_sd + 0.01), (ac_d[:6], ac_h[:6])
    # nothing beyond the 4-sample cell structure: the estimate of a zero correlation from n/4 impulses has sigma 1/sqrt(n/4)
    assert np.all(np.abs(ac_d[4:]) < 4.5 / np.sqrt(len(dev) / 4))


def test_philox_synthesis_noise_is_standard_normal_and_white():
    """The noise of the pulse-wise decode's default path (on-device Philox + Box-Muller; bench.py times this path):
    decode an all-unvoiced encoding with a flat spectrum and unit aperiodicity, so that the output is the noise itself
    shaped only by the (flat) minimum-phase response — then check mean, variance ratio between seeds, whiteness."""
    from world.batch import BatchEncoding, WorldBatch

    fs, nf, k = 16000, 401, 513
    wb = WorldBatch()
    dat = {"f0": np.zeros(nf), "vuv": np.zeros(nf), "temporal_positions": np.arange(nf) * 0.005,
           "spectrogram": np.ones((k, nf)), "aperiodicity": np.full((k, nf), 1 - 1e-12), "fs": fs, "is_requiem": False}
    enc = BatchEncoding.from_dicts(wb.rt, [dat])
    ys = [wb.decode_device(enc, seed=s)[0].cpu().numpy() for s in (1, 2)]
    for y in ys:
        body = y[2000:-2000]
        assert abs(body.mean()) < 0.02 * body.std()
        blocks = body[:len(body) // 32 * 32].reshape(32, -1)
        psd = np.mean(np.abs(np.fft.rfft(blocks, axis=1)) ** 2, axis=0)[2:-2]
        flat = np.exp(np.mean(np.log(psd))) / np.mean(psd)
        assert flat > 0.9, flat  # a white sequence averaged over 32 blocks gives ~0.98
        z = (body - body.mean()) / body.std()
        assert abs(np.mean(z ** 3)) < 0.1 and abs(np.mean(z ** 4) - 3.0) < 0.3  # Gaussian moments
    assert abs(ys[0].std() / ys[1].std() - 1) < 0.05
    assert abs(np.corrcoef(ys[0][2000:-2000], ys[1][2000:-2000])[0, 1]) < 0.02  # seeds are independent streams
