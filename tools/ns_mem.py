"""GPU: device memory one pipeline of the north-star batch (1024 x 10 s, Harvest + Requiem encode + decode) holds, and the
step time with one and two such pipelines in flight.  usage: python tools/ns_mem.py [utterances]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch
from world._synthetic import synth_utterance
from world.batch import WorldBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fs = 16000
base = [synth_utterance(u, fs, 10.0) for u in range(64)]
xs = [base[i % 64] for i in range(n)]
gb = lambda b: round(b / 2 ** 30, 1)
free0, total = torch.cuda.mem_get_info()
print("free at start", gb(free0), "of", gb(total))
wbs, res = [], []
def step(w, r):
    enc = w.encode_device(r[0], r[1], r[2], fs, f0_method="harvest", is_requiem=True, check=False)
    return w.decode_device(enc, check=False)
for d in range(2):
    w = WorldBatch(0, lane=d + 1)
    r = w.upload(xs, fs)
    step(w, r); torch.cuda.synchronize(); w.check("warm")
    wbs.append(w); res.append(r)
    free, _ = torch.cuda.mem_get_info()
    print("pipelines", d + 1, "used GB", gb(free0 - free), "free", gb(free))
    if d == 0 and free < 1.25 * (free0 - free):
        print("no room for a second pipeline"); break
def timed(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6):
        step(wbs[i % k], res[i % k])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 6 * 1e3
for k in range(1, len(wbs) + 1):
    timed(k); print("in flight", k, "ms/step %.2f %.2f" % (timed(k), timed(k)))
