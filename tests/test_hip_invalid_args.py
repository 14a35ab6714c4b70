"""GPU: forty calls with arguments no analysis can be made of (tools/invalid_args_probe.py, in a subprocess with a time limit):
empty / one-sample / 2-D waveforms, rates of 0, -16000, 1000 and 1e9, frame periods of 0 and -5, inverted and zero search
ranges, transform lengths that are no power of two or out of range, dicts with missing, transposed, short or mixed entries,
scale factors of 0 and -1.  Each raises a Python exception or returns; nothing crashes the process, nothing hangs, and the
cases that would make a kernel read past an array are refused on the host."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_invalid_arguments_raise_or_return():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "invalid_args_probe.py")], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "PROBE DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    lines = {ln[:44].strip(): ln[44:].strip() for ln in r.stdout.splitlines() if " raised " in ln or " returned " in ln}
    assert len(lines) == 39, sorted(lines)
    for name in ("decode with f0 of another length", "decode with a transposed spectrogram", "decode_batch of mixed rates", "2-D waveform"):
        assert lines[name].startswith("raised ValueError"), (name, lines[name])
    for name in ("empty waveform", "31 samples (harvest)", "fft_size = 1000", "fs = 1000"):
        assert lines[name].startswith("raised WorldHipError"), (name, lines[name])
