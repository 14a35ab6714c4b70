"""GPU: waveforms holding NaN / Inf / 1e300 next to a clean utterance in one batch, through both pipelines (in a subprocess
with a time limit: the point is that every call RETURNS).  The reference raises or returns NaN on such input; here the call
returns, no kernel hangs, and the clean neighbour's results are the ones it gets alone, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nonfinite_input_returns_and_leaves_its_neighbour_alone():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nonfinite_probe.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PROBE DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "returned in" in ln]
    assert len(lines) == 12 and all("clean neighbour finite: True" in ln for ln in lines), r.stdout[-2000:]


def test_neighbour_of_a_nan_utterance_equals_its_solo_result():
    from world._synthetic import synth_utterance
    from world.batch import WorldBatch

    fs = 16000
    clean = synth_utterance(3, fs, 0.4)
    bad = clean.copy()
    bad[2000:2600] = np.nan
    wb = WorldBatch()
    for method, req in (("dio", False), ("harvest", True)):
        solo = wb.encode([clean], fs, f0_method=method, is_requiem=req).to_dicts()[0]
        pair = wb.encode([bad, clean], fs, f0_method=method, is_requiem=req, check=False).to_dicts()[1]
        for k in ("f0", "vuv", "spectrogram", "aperiodicity"):
            assert np.array_equal(solo[k], pair[k]), (method, k)
        wb.rt.take_flags()
