import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, "golden_%s.npz" % name)))

    return load


def rel_rms(a, b):
    import numpy as np

    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2) / max(np.mean(b ** 2), 1e-300)))


_synth_memo = {}


def synth_cached(u, fs, seconds):
    """world._synthetic.synth_utterance(u, fs, seconds), generated once per test session (a 10 s utterance costs ~0.15 s
    of host time, a 60 s one at 48 kHz ~3 s; several GPU test modules use the same ones).  Read-only: copy before editing."""
    from world._synthetic import synth_utterance

    key = (int(u), int(fs), float(seconds))
    if key not in _synth_memo:
        x = synth_utterance(u, fs, seconds)
        x.setflags(write=False)
        _synth_memo[key] = x
    return _synth_memo[key]
