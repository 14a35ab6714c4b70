"""GPU: a batch whose waveforms hold more than 2^31 samples in total (tools/index64_probe.py, in a subprocess: 24 identical
utterances of 90 M samples built on the device, 35 GB at its peak) — DIO + StoneMask on the 5 ms grid (27 M frames) and
CheapTrick + D4C on a 250 ms grid: the last utterance, whose samples lie beyond the 32-bit range of the batch's flat index,
gets exactly the first one's results."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_offsets_beyond_32_bits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "index64_probe.py")], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "PROBE DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "total samples 2.160e+09" in r.stdout
    assert "last == first: True" in r.stdout
    assert "== the first's: True / True; finite: True" in r.stdout
    assert r.stdout.count("flags [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]") == 2
