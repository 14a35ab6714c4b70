"""The parameter sweep of tests/golden/make_golden.py sweep (reference outputs: golden_sweep.npz), tests/test_oracle_sweep.py
(oracle against them, CPU) and tests/test_hip_sweep.py (HIP path against oracle and fixture, GPU): World.encode arguments
and inputs the other fixtures leave at their defaults — search ranges, frame periods that are not whole milliseconds, DIO's
channels / target rate / allowed range, the fft_size override (which moves the F0 floor), rates whose decimation ratio
rounds to 1 or down, an int16-scaled waveform, a length that ends on an overlap-save tile of the Harvest band filters."""


def sweep_cases():
    """(utterance index, fs, seconds or -samples, amplitude, encode kwargs)"""
    return [
        (301, 16000, 0.6, 1.0, dict(f0_method="harvest", f0_floor=50, f0_ceil=500, frame_period=2.5)),
        (302, 16000, 0.6, 1.0, dict(f0_method="dio", channels_in_octave=4, allowed_range=0.2, frame_period=10)),
        (303, 22050, 0.5, 1.0, dict(f0_method="dio", target_fs=8000, f0_floor=60)),
        (304, 48000, 0.4, 1.0, dict(f0_method="harvest", frame_period=12.5, is_requiem=True)),
        (305, 16000, 0.6, 1.0, dict(f0_method="harvest", fft_size=2048)),
        (306, 11025, 0.6, 1.0, dict(f0_method="harvest")),
        (307, 16000, 0.6, 32767.0, dict(f0_method="dio")),
        (308, 44100, 0.4, 1.0, dict(f0_method="dio", fft_size=4096)),
        (309, 16000, -14337, 1.0, dict(f0_method="harvest", is_requiem=True)),
        (310, 8000, 0.6, 1.0, dict(f0_method="dio", frame_period=1)),
        (311, 32000, 0.4, 1.0, dict(f0_method="harvest", f0_ceil=1200)),
        (312, 24000, 0.5, 1.0, dict(f0_method="dio", frame_period=3, is_requiem=True)),
    ]


def sweep_input(synth_utterance, case):
    u, fs, sec, amp, _ = case
    if sec < 0:
        x = synth_utterance(u, fs, (-sec + 100) / fs)[: int(-sec)]
    else:
        x = synth_utterance(u, fs, sec)
    return x * amp
