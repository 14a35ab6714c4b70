"""Batched 16-bit WAV input / output around the resident pipeline, with the reference callers'
conventions (example/prosody.py:12-13,57; test/speed.py:14): x = int16 / (2**15 - 1) on the way in,
(out * 2**15).astype(int16) on the way out.  The conversions run on the device (wh_pcm16_to_f64 / wh_f64_to_pcm16), so
only 2-byte samples cross PCIe."""
import numpy as np


def read_wavs(paths):
    """(fs, [int16 mono arrays]) of a list of WAV files that share one sampling rate."""
    from scipy.io import wavfile

    fs0, out = None, []
    for p in paths:
        fs, x = wavfile.read(str(p))
        if x.ndim != 1 or x.dtype != np.int16:
            raise ValueError("%s: expected 16-bit mono PCM" % p)
        if fs0 is not None and fs != fs0:
            raise ValueError("%s: sampling rate %d differs from %d" % (p, fs, fs0))
        fs0 = fs
        out.append(x)
    return fs0, out


def encode_wavs(paths, world_batch=None, **encode_kw):
    """Read, upload as int16 and encode a list of WAV files: (fs, BatchEncoding)."""
    from .batch import WorldBatch

    wb = world_batch or WorldBatch()
    fs, pcm = read_wavs(paths)
    batch, x_d, tp_d = wb.upload_pcm16(pcm, fs, encode_kw.get("frame_period", 5))
    return fs, wb.encode_device(batch, x_d, tp_d, fs, **encode_kw)


def write_wavs(paths, fs, world_batch, y, y_off):
    """Write decode_device's output as one 16-bit WAV per utterance."""
    from scipy.io import wavfile

    pcm = world_batch.to_pcm16(y, y_off)
    if len(pcm) != len(paths):
        raise ValueError("%d paths for %d utterances" % (len(paths), len(pcm)))
    for p, v in zip(paths, pcm):
        wavfile.write(str(p), int(fs), v)
