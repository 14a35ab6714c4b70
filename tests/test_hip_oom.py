"""GPU: a Harvest batch whose workspace cannot be allocated (tools/oom_probe.py, in a subprocess: 4096 x 10 s, ~420 GB of
scratch on a 288 GB device, the waveforms 5 GB of device zeros) fails with a message — and the context serves the next,
ordinary call.  (It did not: the runtime kept the failed allocation as its "last error" and the next call's launch check
reported it as its own, "wh_batch_create: out of memory"; wh::fail now clears it.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_workspace_that_cannot_be_allocated_fails_once():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oom_probe.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PROBE DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "oversized call raised:" in r.stdout and "out of memory" in r.stdout
    assert "next call: 201 frames" in r.stdout and "finite True" in r.stdout
