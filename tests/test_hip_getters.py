"""GPU: the A-3 facade getters (world/main.py:27-104) against reference output (tests/golden/golden_getters.npz,
written by make_golden.py getters_fixture from the unmodified reference)."""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _x(g):
    from world._synthetic import synth_utterance

    return synth_utterance(int(g["utt"]), int(g["fs"]), float(g["seconds"]))


@pytest.mark.parametrize("method", ["harvest", "dio"])
def test_get_f0(golden, method):
    from world import main

    g = golden("getters")
    tp, f0, vuv = main.World().get_f0(int(g["fs"]), _x(g), f0_method=method)
    assert np.array_equal(tp, g["getf0_%s_tp" % method])
    assert np.array_equal(vuv, g["getf0_%s_vuv" % method])
    assert np.max(np.abs(f0 - g["getf0_%s_f0" % method])) < 1e-6


def test_get_spectrum(golden):
    from world import main

    g = golden("getters")
    out = main.World().get_spectrum(int(g["fs"]), _x(g), f0_method="dio")
    assert sorted(out.keys()) == list(g["getspec_keys"])
    assert out["spectrogram"].shape == tuple(g["getspec_shape"])
    assert out["ps spectrogram"].shape == tuple(g["getspec_ps_shape"])
    assert out["ps spectrogram"].dtype == np.complex128
    assert np.max(np.abs(out["f0"] - g["getspec_f0"])) < 1e-6  # includes the 500 Hz substitutions (Q6)
    assert rel_rms(out["spectrogram"][:16, :16], g["getspec_head"]) < 1e-8
    assert rel_rms(out["spectrogram"].sum(axis=0), g["getspec_colsum"]) < 1e-8
    assert rel_rms(out["spectrogram"].sum(axis=1), g["getspec_rowsum"]) < 1e-8
    ps = out["ps spectrogram"][:, g["getspec_ps_cols"]]
    assert np.sqrt(np.mean(np.abs(ps - g["getspec_ps"]) ** 2) / np.mean(np.abs(g["getspec_ps"]) ** 2)) < 1e-10


def test_encode_w_gvn_f0(golden):
    from world import main

    g = golden("getters")
    fs = int(g["fs"])
    def fresh():  # the call zeroes source['f0'] on unvoiced frames in place (Q6): every call gets its own copy
        return {"f0": g["gvn_src_f0"].copy(), "vuv": g["gvn_src_vuv"].copy(),
                "temporal_positions": g["gvn_src_tp"].copy()}

    out = main.World().encode_w_gvn_f0(fs, _x(g), fresh(), fft_size=int(g["gvn_fft_size"]), is_requiem=False)
    assert sorted(out.keys()) == list(g["gvn_keys"])
    assert np.max(np.abs(out["f0"] - g["gvn_f0"])) < 1e-9
    assert rel_rms(out["spectrogram"][:16, :16], g["gvn_spec_head"]) < 1e-8
    assert rel_rms(out["spectrogram"].sum(axis=0), g["gvn_spec_colsum"]) < 1e-8
    assert np.max(np.abs(out["aperiodicity"][:16, :16] - g["gvn_ap_head"])) < 1e-7
    assert rel_rms(out["aperiodicity"].sum(axis=0), g["gvn_ap_colsum"]) < 1e-8
    assert np.max(np.abs(out["coarse_ap"] - g["gvn_coarse"])) < 1e-6
    # the reference's quirks: no fft_size -> TypeError at the assert; is_requiem -> KeyError('coarse_ap')
    with pytest.raises(TypeError):
        main.World().encode_w_gvn_f0(fs, _x(g), fresh())
    with pytest.raises(KeyError):
        main.World().encode_w_gvn_f0(fs, _x(g), fresh(), fft_size=1024, is_requiem=True)
