"""Probe (GPU): a batch whose waveforms hold more than 2^31 samples in total (24 identical utterances of 90 M samples: 17 GB of
device memory) through DIO + StoneMask — the last utterance, whose samples lie beyond the 32-bit range of the batch's flat
index, must get exactly the first one's contour."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)
import torch
from world import _hip, _tables
from world.batch import WorldBatch
from world.dio import dio_device
from world.stonemask import stonemask_device

fs, n_utt, n = 16000, 24, 90_000_000
wb = WorldBatch()
rt = wb.rt
nf = _tables.frame_count(n, fs, 5)
print("total samples %.3e (2^31 = %.3e), frames per utterance %d" % (n_utt * n, 2 ** 31, nf), flush=True)
one = torch.empty(n, dtype=torch.float64, device=rt.device)
step = 10_000_000
for a in range(0, n, step):  # a 140 Hz harmonic tone with a slow vibrato, built in pieces
    t = torch.arange(a, min(a + step, n), dtype=torch.float64, device=rt.device) / fs
    ph = 2 * np.pi * (140.0 * t + 3.0 * torch.sin(2 * np.pi * 0.7 * t))
    one[a:a + len(t)] = 0.3 * torch.sin(ph) + 0.1 * torch.sin(2 * ph) + 0.05 * torch.sin(3 * ph)
x_d = one.repeat(n_utt)
del one
batch = rt.make_batch(np.arange(n_utt + 1, dtype=np.int64) * n, np.arange(n_utt + 1, dtype=np.int64) * nf)
tp_d = rt.to_device(np.tile(_tables.frame_times(nf, 5), n_utt))
t0 = time.time()
try:
    f0_d, vuv_d, _, _ = dio_device(rt, batch, x_d, tp_d, fs, 71, 800, 2, 4000, 5, 0.1)
    f0_d = stonemask_device(rt, batch, x_d, tp_d, f0_d, fs, 71)
    torch.cuda.synchronize()
    print("flags", rt.take_flags(), "in %.1f s" % (time.time() - t0))
    f0 = f0_d.view(n_utt, nf)
    vuv = vuv_d.view(n_utt, nf)
    same = bool(torch.equal(f0[0], f0[-1]) and torch.equal(vuv[0], vuv[-1]) and torch.equal(f0[0], f0[n_utt // 2]))
    print("voiced frames of utterance 0: %d of %d; mean f0 %.2f; last == first: %s" % (int(vuv[0].sum()), nf, float(f0[0][vuv[0] > 0].mean()), same))
    # the dense stages and both decodes on a 250 ms frame grid (22 501 frames per utterance: their windows gather samples at
    # flat offsets up to 2.16e9; the spectrogram of the 5 ms grid would be 110 GB)
    from world.batch import WorldBatch as _WB
    period = 250
    nf2 = _tables.frame_count(n, fs, period)
    batch2 = rt.make_batch(np.arange(n_utt + 1, dtype=np.int64) * n, np.arange(n_utt + 1, dtype=np.int64) * nf2)
    tp_h = np.tile(_tables.frame_times(nf2, period), n_utt)
    tp2 = rt.to_device(tp_h)
    batch2.tp_d, batch2.tp_host = tp2, tp_h
    t0 = time.time()
    enc = wb.encode_device(batch2, x_d, tp2, fs, f0_method="dio", frame_period=period, check=False)
    torch.cuda.synchronize()
    print("dense stages: flags", rt.take_flags(), "in %.1f s" % (time.time() - t0))
    sp = enc.spectrogram.view(n_utt, nf2, -1)
    ap = enc.aperiodicity.view(n_utt, nf2, -1)
    print("spectrogram / aperiodicity of the last utterance == the first's: %s / %s; finite: %s" % (
        bool(torch.equal(sp[0], sp[-1])), bool(torch.equal(ap[0], ap[-1])), bool(torch.isfinite(sp).all() and torch.isfinite(ap).all())))
except _hip.WorldHipError as e:
    print("raised:", str(e)[:200])
print("PROBE DONE")
