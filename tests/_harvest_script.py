"""Helper of tests/test_hip_harvest_rounds.py: Harvest of a small ragged batch in THIS process's library configuration
(environment switches such as WH_HV_ITEM_CAP_RT are read once per process); results to an .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def inputs():
    from world._synthetic import synth_utterance

    fs = 16000
    xs = [synth_utterance(120, fs, 1.3), synth_utterance(121, fs, 0.45), synth_utterance(122, fs, 2.1)]
    quiet = synth_utterance(123, fs, 1.6).copy()  # a stretch 60 dB down: candidates from the noise floor alone
    quiet[9000:17000] *= 1e-3
    xs.append(quiet)
    return fs, xs


def inputs_22k():
    from scipy.io import wavfile

    from world._synthetic import synth_utterance

    fs, xi = wavfile.read(os.path.join(ROOT, "tests", "golden", "test-mwm.wav"))
    return int(fs), [xi / (2 ** 15 - 1), synth_utterance(124, int(fs), 0.9)]


def main(out):
    from world import _hip
    from world.harvest import harvest_device
    from world.batch import WorldBatch

    fs, xs = inputs()
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, fs)
    wb.rt.profile(True)
    f0_d, vuv_d = harvest_device(wb.rt, batch, x_d, tp_d, fs, 71, 800, 5)
    prof = dict()
    for name, ms in wb.rt.profile_collect():
        prof[name] = prof.get(name, 0.0) + ms
    wb.rt.profile(False)
    assert wb.rt.take_flags() == [0] * 16
    # a second batch at 22.05 kHz (decimated rate 7350 Hz: 7.35 samples per 1 ms frame, tile edges between samples) with
    # another f0 floor (longer filters, wider margins): the reference's test recording and a synthetic utterance
    fs2, xs2 = inputs_22k()
    batch2, x2_d, tp2_d = wb.upload(xs2, fs2)
    f0_2, vuv_2 = harvest_device(wb.rt, batch2, x2_d, tp2_d, fs2, 60, 700, 5)
    assert wb.rt.take_flags() == [0] * 16
    # the [channel][frame] raw-candidate map of one utterance of each rate (the debug read-out of wh_harvest): the fused
    # and the paired kernels must fill it with the same values
    raws = []
    for fs_r, x_r, args_r in ((fs, xs[3], (71, 800)), (fs2, xs2[0], (60, 700))):
        b1, x1, t1 = wb.upload([x_r], fs_r)
        _, _, dbg = harvest_device(wb.rt, b1, x1, t1, fs_r, args_r[0], args_r[1], 5, debug=True)
        raws.append(dbg["raw"].cpu().numpy())
    assert wb.rt.take_flags() == [0] * 16
    np.savez(out, f0=f0_d.cpu().numpy(), vuv=vuv_d.cpu().numpy(), frame_off=batch.frame_off, raw_16k=raws[0], raw_22k=raws[1],
             f0_22k=f0_2.cpu().numpy(), vuv_22k=vuv_2.cpu().numpy(), frame_off_22k=batch2.frame_off,
             kernels=np.array(sorted(prof)), ms=np.array([prof[k] for k in sorted(prof)]))


if __name__ == "__main__":
    main(sys.argv[1])
