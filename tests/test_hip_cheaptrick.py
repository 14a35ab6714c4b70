"""GPU parity: wh_cheaptrick (through the Python mirror) vs the NumPy oracle and the golden fixture."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["syn16k", "syn48k"])
def test_cheaptrick_vs_golden_and_oracle(golden, tag):
    from oracle import envelope
    from world.cheaptrick import cheaptrick

    g = golden(tag)
    fs = int(g["fs"])
    src = {"f0": g["stonemask_f0"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy()}
    out = cheaptrick(g["x"], fs, src)
    assert out["spectrogram"].shape == g["ct_spectrogram"].shape
    assert np.array_equal(src["f0"], g["ct_f0_after"])  # in-place 500 Hz substitution (Q6)
    # tolerance: north_star 1e-4 relative RMS; FP64 kernels land ~1e-12
    assert rel_rms(out["spectrogram"], g["ct_spectrogram"]) < 1e-9
    sp, ps, _ = envelope.cheaptrick_np(g["x"], fs, g["stonemask_f0"], g["dio_vuv"], g["tp"])
    assert rel_rms(out["spectrogram"], sp) < 1e-9
    assert np.max(np.abs(out["spectrogram"] - sp) / sp) < 1e-4
    assert np.max(np.abs(out["ps spectrogram"] - ps)) < 1e-12
