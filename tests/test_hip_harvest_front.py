"""GPU: the fused Harvest front end (csrc/wh_harvest_front.h, opt-in: WH_HV_FRONT=1) against the default chain and the
oracle.  The switch is read once per process, so every configuration runs in its own subprocess:
  default            band_events -> edge lists -> hv_raw (what ships)
  fused              hv_front_kernel with its standard margins
  fused, margin 0.05 margins so small that most channels of every utterance fail the coverage check and go back through
                     the unfused chain, per channel — the hand-back path, exercised on purpose
All three must give the same voicing decisions and the same f0 (the filtered tiles differ by rounding only: the block
origins differ) and match the oracle's harvest (world/harvest.py:17-54)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, **env):
    out = str(tmp_path / ("front_%s.npz" % tag))
    e = dict(os.environ)
    for k in ("WH_HV_FRONT", "WH_HV_FRONT_MARGIN", "WH_HV_ITEM_CAP_RT"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_front_script.py"), out], capture_output=True, text=True,
                       timeout=600, env=e)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = dict(np.load(out))
    d["prof"] = dict(zip([str(k) for k in d["kernels"]], d["ms"]))
    return d


def test_fused_front_end_equals_the_chain_and_the_oracle(tmp_path):
    from oracle import pitch_harvest
    from _front_script import inputs, inputs_22k

    base = _run(tmp_path, "default")
    fused = _run(tmp_path, "fused", WH_HV_FRONT="1")
    forced = _run(tmp_path, "forced", WH_HV_FRONT="1", WH_HV_FRONT_MARGIN="0.05")
    assert "hv_front_kernel" not in base["prof"] and "hv_front_kernel" in fused["prof"]
    # the hand-back path really ran in the forced configuration: its gated kernels did work there and (next to) none before
    # (a gated kernel with nothing to do is a few microseconds of launch; with work it is tens)
    assert forced["prof"]["hv_raw_kernel"] > 2 * fused["prof"]["hv_raw_kernel"]
    assert forced["prof"]["band_events_kernel"] > 2 * fused["prof"]["band_events_kernel"]
    for other in (fused, forced):
        assert np.array_equal(other["vuv"], base["vuv"])
        assert np.max(np.abs(other["f0"] - base["f0"])) < 1e-7
        assert np.array_equal(other["vuv_22k"], base["vuv_22k"])  # 22.05 kHz (7350 Hz decimated), f0 floor 60 Hz
        assert np.max(np.abs(other["f0_22k"] - base["f0_22k"])) < 1e-7
    fs2, xs2 = inputs_22k()
    fo2 = base["frame_off_22k"]
    for u, x in enumerate(xs2):
        ref = pitch_harvest.harvest_np(x, fs2, 60, 700)
        a, b = int(fo2[u]), int(fo2[u + 1])
        assert np.array_equal(fused["vuv_22k"][a:b], ref["vuv"])
        assert np.max(np.abs(fused["f0_22k"][a:b] - ref["f0"])) < 1e-6
    fs, xs = inputs()
    fo = base["frame_off"]
    for u, x in enumerate(xs):
        ref = pitch_harvest.harvest_np(x, fs)
        a, b = int(fo[u]), int(fo[u + 1])
        for d in (base, fused):
            assert np.array_equal(d["vuv"][a:b], ref["vuv"])
            assert np.max(np.abs(d["f0"][a:b] - ref["f0"])) < 1e-6


def test_refinement_in_several_rounds_equals_one_round(tmp_path):
    """hv_refine_kernel takes the candidates of a block's 24 frames in rounds of whole frames that fit its LDS work list
    (1344 slots: one round on real input).  With the list capped at 120 slots (WH_HV_ITEM_CAP_RT) every block needs
    several rounds, the later ones fetching their rows again: the contour must not change by a bit."""
    base = _run(tmp_path, "default")
    capped = _run(tmp_path, "capped", WH_HV_ITEM_CAP_RT="120")
    assert np.array_equal(capped["vuv"], base["vuv"]) and np.array_equal(capped["f0"], base["f0"])
