"""Probe (GPU): a Harvest batch whose workspace cannot be allocated (4096 x 10 s: ~420 GB of scratch on a 288 GB device; the
waveforms themselves are 5 GB of device zeros) — the call must fail with an error message, and the context must serve the next,
ordinary call."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-world_amd")):
    sys.path.insert(0, p)
import torch
from world import _hip, _tables
from world._synthetic import synth_utterance
from world.batch import WorldBatch
from world.harvest import harvest_device

fs, n_utt, n = 16000, 4096, 160000
wb = WorldBatch()
rt = wb.rt
nf = _tables.frame_count(n, fs, 5)
free_s, total_s = torch.cuda.mem_get_info()
print("device memory free at start: %.1f of %.1f GB" % (free_s / 2 ** 30, total_s / 2 ** 30), flush=True)
batch = rt.make_batch(np.arange(n_utt + 1) * n, np.arange(n_utt + 1) * nf)
x_d = torch.zeros(n_utt * n, dtype=torch.float64, device=rt.device)
tp_d = rt.to_device(np.tile(_tables.frame_times(nf, 5), n_utt))
free0, total = torch.cuda.mem_get_info()
print("device memory free before: %.1f of %.1f GB" % (free0 / 2 ** 30, total / 2 ** 30), flush=True)
try:
    harvest_device(rt, batch, x_d, tp_d, fs)
    torch.cuda.synchronize()
    print("the oversized call RETURNED (device larger than expected?)", rt.take_flags())
except _hip.WorldHipError as e:
    print("oversized call raised:", str(e)[:160])
del x_d, tp_d, batch
torch.cuda.empty_cache()
x = synth_utterance(1, fs, 1.0)
enc = wb.encode([x], fs, f0_method="harvest")
d = enc.to_dicts()[0]
free1, _ = torch.cuda.mem_get_info()
print("next call: %d frames, %d voiced, finite %s; device memory free after: %.1f GB" % (len(d["f0"]), int(d["vuv"].sum()), bool(np.isfinite(d["spectrogram"]).all()), free1 / 2 ** 30))
print("PROBE DONE")
