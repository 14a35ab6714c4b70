"""GPU: a fixed sample of the randomised differential campaign (tools/differential_campaign.py, the first 24 draws of seed 9;
profiles/r06_differential_campaign.txt has the 1600-case runs): rates from 8 to 96 kHz, the three estimators, D4C / D4C-Requiem,
frame periods, search ranges, the fft_size override, amplitudes, zero padding, noise, DC, clipping, and scale_pitch /
scale_duration before the decode — judged by the campaign's own rule: frame times and VUV exact, f0 1e-6, spectrogram 1e-6,
aperiodicity 1e-5, decode 1e-8 against the oracle (world/main.py:106-214)."""
import os
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_first_draws_of_the_campaign_against_the_oracle():
    import differential_campaign as dc
    from oracle import api as oapi
    from world.batch import WorldBatch
    from world.get_seeds_signals import get_seeds_signals

    wb = WorldBatch()
    seeds_by_fs = {}
    methods, rates = set(), set()
    for i in range(24):
        c = dc.draw_case(i, 9)
        x = dc.make_input(c)
        methods.add(c["kw"]["f0_method"])
        rates.add(c["fs"])
        enc = wb.encode([x], c["fs"], **c["kw"])
        d = enc.to_dicts()[0]
        o = dc.oracle_encode(c)
        assert o["ok"], (c, o.get("error"))
        if c["shape"]["scale_pitch"]:
            enc.scale_pitch(c["shape"]["scale_pitch"])
        if c["shape"]["scale_duration"]:
            enc.scale_duration(c["shape"]["scale_duration"])
        dm = enc.to_dicts()[0]
        if c["kw"]["is_requiem"]:
            if c["fs"] not in seeds_by_fs:
                random.seed(7)
                np.random.seed(7)
                seeds_by_fs[c["fs"]] = get_seeds_signals(c["fs"])
            y, _ = wb.decode_device(enc, seeds=seeds_by_fs[c["fs"]])
            yo = oapi.decode_np(dict(dm), seeds=seeds_by_fs[c["fs"]])["out"]
        else:
            stretch = max(1.0, c["shape"]["scale_duration"] or 1.0)
            noise = np.random.RandomState(c["noise_seed"] + 1).randn(int(2 * len(x) * stretch) + 16384)
            y, _ = wb.decode_device(enc, noise=[noise])
            yo = oapi.decode_np(dict(dm), noise=noise)["out"]
        row = dc.compare_case(c, d, o, y.cpu().numpy(), yo)
        assert not row["fail"], {k: v for k, v in row.items() if k != "shape"}
    assert methods == {"dio", "harvest", "swipe"} and len(rates) >= 8
    assert wb.rt.take_flags() == [0] * 16
