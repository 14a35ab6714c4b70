"""Deterministic synthetic speech-like input for the benchmark configs.

Not part of the reference API.  Utterance ``u`` is a vibrato harmonic source gated 0.8 s voiced /
0.2 s unvoiced, passed through three cascaded formant resonators, peak-normalised to 0.5, with a
1e-3 white floor added afterwards (without the floor the reference's D4C divides by an exactly-zero
smoothed power and produces NaN — SURVEY §7.3 Q14).
"""
import numpy as np


def _resonator(sig: np.ndarray, fc: float, bw: float, fs: float) -> np.ndarray:
    r = np.exp(-np.pi * bw / fs)
    theta = 2.0 * np.pi * fc / fs
    a1 = -2.0 * r * np.cos(theta)
    a2 = r * r
    b0 = 1.0 - r
    from scipy.signal import lfilter

    return lfilter([b0], [1.0, a1, a2], sig)


def synth_utterance(u: int, fs: int = 16000, seconds: float = 10.0) -> np.ndarray:
    """float64 waveform of ``int(fs*seconds)`` samples for utterance index ``u``."""
    n = int(round(fs * seconds))
    rng = np.random.RandomState(1234 + u)
    t = np.arange(n) / fs
    base = 90.0 + 160.0 * rng.rand()
    rate = 0.3 + 0.4 * rng.rand()
    ph0 = 2.0 * np.pi * rng.rand()
    f0 = base * 2.0 ** (0.25 * np.sin(2.0 * np.pi * rate * t + ph0))
    voiced = np.mod(t + rng.rand(), 1.0) < 0.8
    phi = 2.0 * np.pi * np.cumsum(f0) / fs
    n_harm = int((fs / 2) // f0.max())
    src = np.zeros(n)
    for k in range(1, n_harm + 1):
        src += np.cos(k * phi) / k
    noise = 0.3 * rng.randn(n)
    sig = np.where(voiced, src, noise)
    for lo, span, bw in ((500.0, 300.0, 80.0), (1500.0, 500.0, 120.0), (2500.0, 500.0, 160.0)):
        sig = _resonator(sig, lo + span * rng.rand(), bw, fs)
    sig = 0.5 * sig / np.max(np.abs(sig))
    sig = sig + 1e-3 * rng.randn(n)
    return sig


def synth_batch(count: int, fs: int = 16000, seconds: float = 10.0, first: int = 0) -> np.ndarray:
    """(count, N) float64 batch, utterance indices first … first+count-1."""
    return np.stack([synth_utterance(first + i, fs, seconds) for i in range(count)])


def harmonic_tone(fs: int, f0: float, seconds: float = 0.8, vibrato: float = 0.03) -> np.ndarray:
    """Harmonic tone (1/k partials up to 0.45 fs) whose pitch sweeps f0 x (1 +- vibrato) at 2.5 Hz, with a 1e-3 white
    floor; seeded by (fs, f0).  Test input for pitch estimators: the sweep walks the argmax over the candidates
    around f0 (tests/golden/make_golden.py swipe_fixture)."""
    t = np.arange(int(round(fs * seconds))) / fs
    phi = 2.0 * np.pi * np.cumsum(f0 * (1.0 + vibrato * np.sin(2.0 * np.pi * 2.5 * t))) / fs
    x = np.zeros(len(t))
    for k in range(1, int(0.45 * fs // (f0 * (1.0 + vibrato))) + 1):
        x += np.cos(k * phi + 0.37 * k * k) / k
    x = 0.4 * x / np.max(np.abs(x))
    return x + 1e-3 * np.random.RandomState(int(fs + f0)).randn(len(t))
