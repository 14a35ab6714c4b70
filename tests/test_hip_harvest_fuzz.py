"""GPU: Harvest on signals outside the speech-like regime, both forms of the raw-candidate stage (the fused
hv_rawdet_kernel, forced with WH_HV_RAWDET_MIN_TILES=0, and the hv_raw / hv_detect pair), each in its own process
(tests/_harvest_script.py fuzz), against the oracle (world/harvest.py:17-54): stretches of digital silence between tone
bursts (event lists with long gaps: the cursor hints of the fused kernel point into nothing there), white noise (every
channel carries a candidate), a chirp across the whole search range, a click train, a DC offset, a 0.2 s and a 1e-8
amplitude utterance, two tones, digital silence.  VUV exact, f0 to 1e-6 Hz.

The zero-crossing lists (the reference's ragged arrays, world/harvest.py:283-297): digital silence next to signal
becomes a DC level under harvest.py:69's mean removal, the first difference of its filtered image changes sign at
random, and the lists outgrow their estimated capacity — WH_FLAG_EVENT_OVERFLOW on the first pass, none on the repeat
with the counted capacities (wh_harvest_event_counts / wh_harvest_set_event_caps) nor with event_caps='safe', the same
bits from both; and the facade (World.encode_batch, World.get_f0) does that repeat by itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, **env):
    out = str(tmp_path / ("fuzz_%s.npz" % tag))
    e = dict(os.environ)
    e.pop("WH_HV_RAWDET_MIN_TILES", None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_harvest_script.py"), out, "fuzz"], capture_output=True,
                       text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return dict(np.load(out))


def test_harvest_off_regime_signals_both_forms_against_the_oracle(tmp_path):
    from oracle import pitch_harvest
    from _harvest_script import fuzz_inputs

    fused = _run(tmp_path, "fused", WH_HV_RAWDET_MIN_TILES="0")
    paired = _run(tmp_path, "paired", WH_HV_RAWDET_MIN_TILES="1000000000")
    ov = int(fused["overflow_flag"])
    for r in (fused, paired):
        first = list(r["flags_first"])
        assert first[ov] == 1 and sum(first) == 1  # the estimate is exceeded (the bursts between digital silence) ...
        assert list(r["flags"]) == [0] * 16        # ... the counted capacities hold the same crossings,
        assert np.array_equal(r["caps"], r["caps_again"])
        assert list(r["flags_safe"]) == [0] * 16   # and so does the bound — with the same result, bit for bit
        assert np.array_equal(r["vuv"], r["vuv_safe"]) and np.array_equal(r["f0"], r["f0_safe"])
        assert list(r["flags_after"]) == first     # ('safe' served its own call only)
    assert np.array_equal(fused["vuv"], paired["vuv"])
    assert np.max(np.abs(fused["f0"] - paired["f0"])) < 1e-9
    fs, xs = fuzz_inputs()
    fo = fused["frame_off"]
    voiced_total = 0
    for u, x in enumerate(xs):
        a, b = int(fo[u]), int(fo[u + 1])
        if not np.any(x):  # the reference divides by zero on an all-zero signal; this build: unvoiced
            assert not fused["vuv"][a:b].any() and not fused["f0"][a:b].any()
            continue
        ref = pitch_harvest.harvest_np(x, fs, 71, 800, 5, return_aux=True)
        assert np.array_equal(fused["vuv"][a:b], ref["vuv"]), u
        assert np.max(np.abs(fused["f0"][a:b] - ref["f0"])) < 1e-6, u
        voiced_total += int(ref["vuv"].sum())
        # stage by stage: the 1 ms contour everywhere; the raw [channel][frame] candidates where the reference's own are
        # not rounding noise — next to digital silence (signal 0) the filtered DC level's first difference is, in its
        # arithmetic as in this one, and a sixth of the map's entries differ between the two without the contour noticing
        aux = ref["aux"]
        for r in (fused, paired):
            f1 = r["f1_%d" % u][: len(aux["f0_1ms"])]
            assert np.array_equal(f1 != 0, aux["f0_1ms"] != 0), u
            assert np.max(np.abs(f1 - aux["f0_1ms"])) < 1e-6, u
            if u == 0:
                continue
            nb, nf1 = aux["raw"].shape
            raw = r["raw_%d" % u][: nb * nf1].reshape(nb, nf1)
            assert np.array_equal(raw != 0, aux["raw"] != 0), u
            assert np.max(np.abs(raw - aux["raw"])) < 1e-6, u
    assert voiced_total > 200  # (the bursts, the chirp, the clicks and the tones are voiced somewhere)


def test_facade_repeats_a_harvest_whose_event_lists_overflowed():
    """World.get_f0 (one utterance), World.encode_batch (one batch, and the two-part form forced by a tiny split threshold)
    and the thread-per-device pool on the burst signal: no error, and the contour of the explicit repeat."""
    from _harvest_script import fuzz_inputs
    from world import main as wmain
    from world.batch import WorldBatch
    from world.harvest import counted_event_caps, harvest_device
    from world import _hip

    fs, xs = fuzz_inputs()
    bursts, chirp = xs[0], xs[2]
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload([bursts], fs)
    harvest_device(wb.rt, batch, x_d, tp_d, fs)
    assert wb.rt.take_flags()[_hip.FLAG_EVENT_OVERFLOW] == 1
    f0_d, vuv_d = harvest_device(wb.rt, batch, x_d, tp_d, fs, event_caps=counted_event_caps(wb.rt))
    assert wb.rt.take_flags() == [0] * 16
    f0_ref, vuv_ref = f0_d.cpu().numpy(), vuv_d.cpu().numpy()
    w = wmain.World()
    _, f0, vuv = w.get_f0(fs, bursts)
    assert np.array_equal(f0, f0_ref) and np.array_equal(vuv, vuv_ref)
    dats = w.encode_batch(fs, [chirp, bursts, chirp])
    assert np.array_equal(dats[1]['f0'][dats[1]['vuv'] > 0], f0_ref[vuv_ref > 0]) and np.array_equal(dats[1]['vuv'], vuv_ref)
    split = wmain.FACADE_SPLIT_BYTES
    try:
        wmain.FACADE_SPLIT_BYTES = 1
        parts = w.encode_batch(fs, [chirp, bursts, chirp])
    finally:
        wmain.FACADE_SPLIT_BYTES = split
    pooled = w.encode_batch(fs, [chirp, bursts, chirp], devices=[0, 0])
    for other in (parts, pooled):
        for a, b in zip(dats, other):
            assert np.array_equal(a['f0'], b['f0']) and np.array_equal(a['vuv'], b['vuv'])
            assert np.array_equal(a['spectrogram'], b['spectrogram'])
    assert _hip.Runtime.get().take_flags() == [0] * 16


def test_zero_padded_utterance_fits_the_hinted_capacities_and_matches_the_oracle():
    """A speech-like utterance between 0.4 s of digital silence (a zero-padded clip).  Sized from the length alone the
    crossing lists overflow; an encode that has seen the host array (WorldBatch.encode, World.encode) sizes them from
    its flat samples as well and needs no repeat; the contour is the oracle's (VUV exact, f0 to 1e-6 Hz)."""
    import torch

    from oracle import pitch_harvest
    from world import _hip
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch
    from world.harvest import counted_event_caps, harvest_device

    fs = 16000
    pad = np.zeros(int(0.4 * fs))
    x = np.concatenate([pad, synth_utterance(131, fs, 2.0), pad])
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload([x], fs)
    harvest_device(wb.rt, batch, x_d, tp_d, fs)
    assert wb.rt.take_flags()[_hip.FLAG_EVENT_OVERFLOW] == 1  # from the length alone: exceeded
    f0_c, vuv_c = harvest_device(wb.rt, batch, x_d, tp_d, fs, event_caps=counted_event_caps(wb.rt))
    assert wb.rt.take_flags() == [0] * 16
    enc = wb.encode([x], fs, f0_method='harvest', check=False)
    torch.cuda.synchronize()
    assert wb.rt.take_flags() == [0] * 16  # with the flat samples counted at upload: fits, first time
    assert torch.equal(enc.vuv, vuv_c)
    ref = pitch_harvest.harvest_np(x, fs, 71, 800, 5)
    assert np.array_equal(vuv_c.cpu().numpy(), ref["vuv"])
    assert np.max(np.abs(f0_c.cpu().numpy() - ref["f0"])) < 1e-6
    assert ref["vuv"].sum() > 100 and not ref["vuv"][:60].any()
