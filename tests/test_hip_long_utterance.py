"""GPU: single very long utterances through the drop-in facade (tools/long_utterance_probe.py, in a subprocess): 30 minutes at
16 kHz on both pipelines and 10 minutes at 48 kHz — no flag, finite output of the reference's float-arange length, the same
bits on a second run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_half_hour_utterances():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "long_utterance_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBE DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if " min at " in ln]
    assert len(lines) == 3
    for ln in lines:
        assert "finite True, second run identical True, flags []" in ln, ln
        assert "decode 28800001 samples (expected 28800001)" in ln, ln
