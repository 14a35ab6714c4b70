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


if __name__ == "__main__" and len(sys.argv) == 2:
    main(sys.argv[1])


def fuzz_inputs(fs=16000):
    """Signals that leave the speech-like regime (tests/test_hip_harvest_fuzz.py): gaps without a single crossing, noise,
    a chirp through the whole search range, clicks, a DC offset, a very short and a very quiet utterance, two tones.
    At another rate than 16 kHz the tones are detuned by 0.3 % (no whole number of decimated samples per half period
    at 7350 or 8000 Hz, see the click train below)."""
    k = 1.0 if fs == 16000 else 1.00317
    rng = np.random.RandomState(77)
    t = np.arange(int(1.2 * fs)) / fs
    out = []
    x = np.zeros_like(t)  # tone bursts between stretches of digital silence
    for a, b, f in ((0.10, 0.35, 120.0 * k), (0.55, 0.70, 310.0 * k), (0.95, 1.15, 75.0 * k)):
        m = (t >= a) & (t < b)
        x[m] = 0.3 * np.sin(2 * np.pi * f * t[m])
    out.append(x)
    out.append(0.1 * rng.randn(len(t)))  # white noise: every channel "live"
    ph = 2 * np.pi * np.cumsum(np.linspace(60.0, 900.0, len(t))) / fs
    out.append(0.4 * np.sin(ph) + 0.1 * np.sin(3 * ph))  # chirp across the search range
    # click train at 140.7 Hz.  (Not at a whole number of decimated samples per period: a dip of the filtered train then
    # lies exactly midway between two samples, their difference is rounding noise, and where it comes out as an exact 0
    # neither neighbouring product d0*d1 is negative — ZeroCrossingEngine, harvest.py:283-297, misses that dip, the
    # interval doubles, and which dips go that way is the FFT library's rounding: 19 231 of 70 291 raw candidates differed
    # between two correct implementations.  The same goes for a tone with a whole number of samples per half period.
    # Amplitudes differ from click to click for the same reason: equal clicks at mirrored distances make the filtered
    # signal exactly symmetric about a click, and a click on an odd sample sits between two decimated samples.)
    x = np.zeros_like(t)
    at = np.round(np.arange(0, len(t) - 3, fs / 140.7)).astype(int)
    x[at] = 0.6 + 0.4 * rng.rand(len(at))
    x[at + 1] = 0.35 * x[at]  # (and a click is three unequal samples: a single one filters into a response symmetric about itself,
    x[at + 2] = -0.2 * x[at]  #  alone in the short windows of the channels above 500 Hz)
    out.append(x)
    out.append(0.5 + 1e-3 * rng.randn(len(t)))  # DC offset + a little noise
    out.append(0.3 * np.sin(2 * np.pi * 203.7 * k * t[: int(0.2 * fs)] + 0.3))  # 0.2 s
    out.append(1e-8 * (np.sin(2 * np.pi * 150.0 * k * t) + 0.01 * rng.randn(len(t))))  # very quiet
    out.append(0.3 * np.sin(2 * np.pi * 110.0 * k * t) + 0.25 * np.sin(2 * np.pi * 173.0 * k * t + 1.0))  # two tones
    out.append(np.zeros(int(0.5 * fs)))  # digital silence (the reference raises; this build returns unvoiced)
    return fs, out


def main_fuzz(out):
    """Harvest of the off-regime signals as one batch.  The estimate of the zero-crossing lists' capacities fails on the
    tone bursts between digital silence (WH_FLAG_EVENT_OVERFLOW); the call is then repeated with the capacities it counted,
    and once more with event_caps='safe': all three flag states and the two results are stored."""
    from world import _hip
    from world.batch import WorldBatch
    from world.harvest import counted_event_caps, harvest_device

    fs, xs = fuzz_inputs()
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, fs)
    f0_d, vuv_d = harvest_device(wb.rt, batch, x_d, tp_d, fs, 71, 800, 5)
    flags_first = wb.rt.take_flags()
    caps = counted_event_caps(wb.rt)
    f0_d, vuv_d = harvest_device(wb.rt, batch, x_d, tp_d, fs, 71, 800, 5, event_caps=caps)
    flags = wb.rt.take_flags()
    caps_again = counted_event_caps(wb.rt)
    f0_s, vuv_s = harvest_device(wb.rt, batch, x_d, tp_d, fs, 71, 800, 5, event_caps='safe')
    flags_safe = wb.rt.take_flags()
    f0_e, vuv_e = harvest_device(wb.rt, batch, x_d, tp_d, fs, 71, 800, 5)  # ('safe' serves its own call only)
    flags_after = wb.rt.take_flags()
    # every signal on its own with the debug read-out: the [channel][1 ms frame] raw-candidate map and the 1 ms contour
    stages = {}
    for u, x in enumerate(xs):
        if not np.any(x):
            continue
        b1, x1, t1 = wb.upload([x], fs)
        _, _, dbg = harvest_device(wb.rt, b1, x1, t1, fs, 71, 800, 5, debug=True)
        if wb.rt.take_flags()[_hip.FLAG_EVENT_OVERFLOW]:
            _, _, dbg = harvest_device(wb.rt, b1, x1, t1, fs, 71, 800, 5, debug=True, event_caps=counted_event_caps(wb.rt))
            assert wb.rt.take_flags() == [0] * 16
        stages["raw_%d" % u] = dbg["raw"].cpu().numpy()
        stages["f1_%d" % u] = dbg["f0_1ms"].cpu().numpy()
    np.savez(out, f0=f0_d.cpu().numpy(), vuv=vuv_d.cpu().numpy(), frame_off=batch.frame_off, flags=np.array(flags),
             flags_first=np.array(flags_first), caps=caps, caps_again=caps_again, f0_safe=f0_s.cpu().numpy(), **stages,
             vuv_safe=vuv_s.cpu().numpy(), flags_safe=np.array(flags_safe), flags_after=np.array(flags_after),
             overflow_flag=_hip.FLAG_EVENT_OVERFLOW)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "fuzz":
    main_fuzz(sys.argv[1])
