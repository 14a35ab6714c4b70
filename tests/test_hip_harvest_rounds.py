"""GPU: Harvest in library configurations that are read once per process (environment switches), each in its own
subprocess (tests/_harvest_script.py), compared with the default configuration and the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, **env):
    out = str(tmp_path / ("harvest_%s.npz" % tag))
    e = dict(os.environ)
    for k in ("WH_HV_ITEM_CAP_RT", "WH_HV_RAWDET_MIN_TILES"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_harvest_script.py"), out], capture_output=True, text=True,
                       timeout=600, env=e)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = dict(np.load(out))
    d["prof"] = dict(zip([str(k) for k in d["kernels"]], d["ms"]))
    return d


def test_refinement_in_several_rounds_equals_one_round(tmp_path):
    """hv_refine_kernel takes the candidates of a block's 24 frames in rounds of whole frames that fit its LDS work list
    (1344 slots: one round on real input).  With the list capped at 120 slots (WH_HV_ITEM_CAP_RT) every block needs
    several rounds, the later ones fetching their rows again: the contour must not change by a bit."""
    from oracle import pitch_harvest
    from _harvest_script import inputs, inputs_22k

    base = _run(tmp_path, "default")
    capped = _run(tmp_path, "capped", WH_HV_ITEM_CAP_RT="120")
    assert np.array_equal(capped["vuv"], base["vuv"]) and np.array_equal(capped["f0"], base["f0"])
    assert np.array_equal(capped["vuv_22k"], base["vuv_22k"]) and np.array_equal(capped["f0_22k"], base["f0_22k"])
    # and both equal the oracle's harvest (world/harvest.py:17-54): 16 kHz batch with a 60 dB quiet stretch; 22.05 kHz
    # (7350 Hz decimated rate, f0 floor 60 Hz)
    for (fs, xs), fo, kf, kv, args in ((inputs(), base["frame_off"], "f0", "vuv", ()),
                                       (inputs_22k(), base["frame_off_22k"], "f0_22k", "vuv_22k", (60, 700))):
        for u, x in enumerate(xs):
            ref = pitch_harvest.harvest_np(x, fs, *args)
            a, b = int(fo[u]), int(fo[u + 1])
            assert np.array_equal(base[kv][a:b], ref["vuv"])
            assert np.max(np.abs(base[kf][a:b] - ref["f0"])) < 1e-6


def test_fused_and_paired_raw_candidate_kernels_agree(tmp_path):
    """Round 6: large batches take hv_rawdet_kernel (a wave per 64-frame tile walks all channels from the band walker's
    cursor hints; harvest.py:88-110 and :252-278 in one pass), small ones hv_raw_kernel + hv_detect_kernel
    (WH_HV_RAWDET_MIN_TILES).  Both forms on the same inputs — a ragged 16 kHz batch with a 60 dB quiet stretch, and a
    22.05 kHz batch at another f0 floor, whose 64-frame tiles start between decimated samples: VUV equal, f0 to 1e-9 (the
    fused kernel adds a run's candidates channel after channel where np.mean adds pairwise), each kernel really ran, and
    both equal the oracle."""
    from oracle import pitch_harvest
    from _harvest_script import inputs, inputs_22k

    fused = _run(tmp_path, "fused", WH_HV_RAWDET_MIN_TILES="0")
    paired = _run(tmp_path, "paired", WH_HV_RAWDET_MIN_TILES="1000000000")
    assert "hv_rawdet_kernel" in fused["prof"] and "hv_raw_kernel" not in fused["prof"]
    assert "hv_raw_kernel" in paired["prof"] and "hv_rawdet_kernel" not in paired["prof"]
    for kf, kv in (("f0", "vuv"), ("f0_22k", "vuv_22k")):
        assert np.array_equal(fused[kv], paired[kv])
        assert np.max(np.abs(fused[kf] - paired[kf])) < 1e-9
    # the raw candidates themselves (harvest.py:252-278), channel by channel and frame by frame: the fused kernel's
    # interpolation is hv_raw_kernel's, value for value — bitwise the same map, quiet stretch and 7350 Hz rate included
    for k in ("raw_16k", "raw_22k"):
        assert fused[k].shape == paired[k].shape and (fused[k] != 0).sum() > 1000
        assert np.array_equal(fused[k], paired[k]), k
    for (fs, xs), fo, kf, kv, args in ((inputs(), fused["frame_off"], "f0", "vuv", ()),
                                       (inputs_22k(), fused["frame_off_22k"], "f0_22k", "vuv_22k", (60, 700))):
        for u, x in enumerate(xs):
            ref = pitch_harvest.harvest_np(x, fs, *args)
            a, b = int(fo[u]), int(fo[u + 1])
            for d in (fused, paired):
                assert np.array_equal(d[kv][a:b], ref["vuv"])
                assert np.max(np.abs(d[kf][a:b] - ref["f0"])) < 1e-6
