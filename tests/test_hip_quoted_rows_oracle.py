"""GPU: one FULL-LENGTH (10 s) row of each quoted-size batch against the ORACLE.

tests/test_hip_quoted_sizes.py / test_hip_fullsize.py compare rows of the big batches with the same utterance processed
alone (bitwise) — a statement about batching, not about the reference.  Here a row from the middle of

  * config 2: 64 x 10 s, DIO + StoneMask + CheapTrick + D4C, pulse-wise decode with the device Philox stream,
  * config 3: 256 x 10 s, Harvest,
  * config 4 / north star: 1024 x 10 s, Harvest + CheapTrick + D4C-Requiem, Requiem decode at the chained noise cursor,

is checked against oracle/ (the NumPy restatement pinned by the reference's fixtures) at the suite's tolerances: frame
times and VUV exact, f0 1e-6 Hz, spectrogram 1e-9 relative RMS, aperiodicity 1e-7 (D4C) / 1e-6 dB (Requiem bands); the
decode of the batch's row against the oracle's synthesis of that row's encoding, fed the noise the device drew
(`wh_philox_normals` dump of utterance r's stream; the device-built Requiem seed tables + the cursor after r utterances):
1e-9.  Reference: world/main.py:106-152,198-214, world/synthesis.py:61-81, world/synthesisRequiem.py:99-141.
The oracle needs ~10 s (DIO path) / ~20 s (Harvest) per 10 s utterance; the Harvest contour of the row's utterance is
computed once for configs 3 and 4."""
import functools

import numpy as np
import pytest

from conftest import rel_rms, synth_cached

pytestmark = pytest.mark.gpu

FS = 16000
NF = 2001
NY = 160001
UTT = 1  # the row's utterance: rows 37 / 129 / 513 of the three batches are all base[... % 64] chosen below


@functools.lru_cache(maxsize=None)
def _oracle_source(u, method):
    from oracle import pitch_dio, pitch_harvest

    x = synth_cached(u, FS, 10.0)
    if method == "harvest":
        return pitch_harvest.harvest_np(x, FS, 71, 800, 5)
    src = pitch_dio.dio_np(x, FS, 71, 800, 2, 4000, 5, 0.1)
    src["f0"] = pitch_dio.stonemask_np(x, FS, src["temporal_positions"], src["f0"])
    return src


def _oracle_encode(u, method, requiem):
    """oracle.api.encode_np with the F0 stage memoised (world/main.py:106-152)."""
    from oracle import aperiodicity, envelope

    x = synth_cached(u, FS, 10.0)
    src = _oracle_source(u, method)
    tp, vuv = src["temporal_positions"], src["vuv"]
    spec, _, f0_ct = envelope.cheaptrick_np(x, FS, src["f0"].copy(), vuv, tp, want_ps=False)
    if requiem:
        ap, f0_out = aperiodicity.d4c_requiem_np(x, FS, f0_ct, vuv, tp)
    else:
        ap, _, f0_out = aperiodicity.d4c_np(x, FS, f0_ct, vuv, tp)
    return {"temporal_positions": tp, "vuv": vuv, "f0": f0_out, "spectrogram": spec, "aperiodicity": ap}


def _row_dict(enc, r):
    """Row r of a resident batch encoding as a reference-layout dict ((bins, frames) arrays): only this row is downloaded."""
    fo = enc.batch.frame_off
    s = slice(int(fo[r]), int(fo[r + 1]))
    return {"temporal_positions": enc.temporal_positions[s].cpu().numpy(), "vuv": enc.vuv[s].cpu().numpy(),
            "f0": enc.f0[s].cpu().numpy(), "fs": enc.fs, "is_requiem": enc.is_requiem,
            "spectrogram": enc.spectrogram[s].transpose(0, 1).contiguous().cpu().numpy(),
            "aperiodicity": enc.aperiodicity[s].transpose(0, 1).contiguous().cpu().numpy()}


def _check_encoding(d, o, requiem):
    assert np.array_equal(d["temporal_positions"], o["temporal_positions"])
    assert np.array_equal(d["vuv"], o["vuv"])
    assert np.max(np.abs(d["f0"] - o["f0"])) < 1e-6
    assert d["spectrogram"].shape == o["spectrogram"].shape and rel_rms(d["spectrogram"], o["spectrogram"]) < 1e-9
    assert d["aperiodicity"].shape == o["aperiodicity"].shape
    assert np.max(np.abs(d["aperiodicity"] - o["aperiodicity"])) < (1e-6 if requiem else 1e-7)


def test_config2_row_of_64x10s_against_the_oracle():
    from oracle import api as oapi
    from world.batch import WorldBatch
    from world.synthesis import philox_normals

    xs = [synth_cached(u, FS, 10.0) for u in range(64)]
    r = 37
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="dio")
    y, y_off = wb.decode_device(enc, seed=7)
    assert wb.rt.take_flags() == [0] * 16
    d = _row_dict(enc, r)
    _check_encoding(d, _oracle_encode(r, "dio", False), False)
    # the decode: the oracle's synthesis of THIS encoding, drawing the stream utterance r of the batch drew
    dump = philox_normals(wb.rt, 7, r, 2 * NY + 64).cpu().numpy()
    yo = oapi.decode_np(dict(d), noise=dump)["out"]
    seg = y[int(y_off[r]):int(y_off[r + 1])].cpu().numpy()
    assert len(seg) == len(yo) == NY
    assert rel_rms(seg, yo) < 1e-9 and np.max(np.abs(seg - yo)) < 1e-9 * max(1.0, np.max(np.abs(yo)))


def test_config3_row_of_256x10s_against_the_oracle():
    from world.batch import WorldBatch
    from world.harvest import harvest_device

    base = [synth_cached(u, FS, 10.0) for u in range(64)]
    xs = base * 4
    r = 128 + UTT
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, FS)
    with wb.rt.on_stream():
        f0, vuv = harvest_device(wb.rt, batch, x_d, tp_d, FS, 71, 800, 5)
    assert wb.rt.take_flags() == [0] * 16
    o = _oracle_source(UTT, "harvest")
    s = slice(r * NF, (r + 1) * NF)
    assert np.array_equal(tp_d[s].cpu().numpy(), o["temporal_positions"])
    assert np.array_equal(vuv[s].cpu().numpy(), o["vuv"])
    assert np.max(np.abs(f0[s].cpu().numpy() - o["f0"])) < 1e-6
    assert 0.5 < float(np.mean(o["vuv"])) < 0.95  # (a contour with both kinds of frames)


def test_north_star_row_of_1024x10s_against_the_oracle():
    from oracle import api as oapi
    from world.batch import WorldBatch
    from world.synthesisRequiem import _advance, _default_seeds

    base = [synth_cached(u, FS, 10.0) for u in range(64)]
    xs = base * 16
    r = 512 + UTT
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="harvest", is_requiem=True)
    y, y_off = wb.decode_device(enc)
    assert wb.rt.take_flags() == [0] * 16
    d = _row_dict(enc, r)
    _check_encoding(d, _oracle_encode(UTT, "harvest", True), True)
    # Requiem decode of the row: the oracle's synthesisRequiem of THIS encoding with the tables the device built and the
    # cursor the r utterances before it leave behind (world/synthesisRequiem.py:131-141)
    tabs = _default_seeds[(FS, wb.rt.index, wb.rt.lane)]
    seeds = {"pulse": tabs["pulse_d"].cpu().numpy(), "noise": tabs["noise_d"].cpu().numpy()}
    cur = np.zeros(seeds["noise"].shape[1])
    for _ in range(r):
        cur = _advance(cur, NY, seeds["noise"].shape[0])
    yo = oapi.decode_np(dict(d), seeds=seeds, cursor=cur)["out"]
    seg = y[int(y_off[r]):int(y_off[r + 1])].cpu().numpy()
    assert len(seg) == len(yo) == NY
    assert rel_rms(seg, yo) < 1e-9 and np.max(np.abs(seg - yo)) < 1e-9 * max(1.0, np.max(np.abs(yo)))
    del y, enc
    wb.rt.torch.cuda.empty_cache()
