#!/usr/bin/env python3
"""Where the drop-in face's resynthesis flow spends its time (bench.py facade_batch: encode_batch -> scale_pitch ->
scale_duration -> decode_batch on 64 x 10 s): upload / encode / to_dicts(lazy) / the modifiers / from_dicts / decode /
download / slicing, four repeats (the first two pay for the pinned staging blocks).  python tools/facade_profile.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "python-world_amd")
import numpy as np, torch
import bench
from world import main
from world.batch import WorldBatch, BatchEncoding
xs = bench.make_inputs(0, 64, 16000, 10.0)
fs = 16000
W = main.World()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = T()
    wb = WorldBatch()
    batch, x_d, tp_d = wb.upload(xs, fs); t1 = T()
    enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio"); t2 = T()
    dats = enc.to_dicts(lazy=True); t3 = T()
    for d in dats:
        W.scale_pitch(d, 1.5); W.scale_duration(d, 2.0)
    t4 = T()
    e2 = BatchEncoding.from_dicts(wb.rt, dats); t5 = T()
    y, y_off = wb.decode_device(e2); t6 = T()
    with wb.rt.on_stream():
        yh = wb.rt.to_host(y)
    t7 = T()
    for u, d in enumerate(dats):
        d['out'] = yh[int(y_off[u]):int(y_off[u + 1])]
    t8 = T()
    print("upload %.2f encode %.2f to_dicts %.2f scale %.2f from_dicts %.2f decode %.2f to_host %.2f assign %.2f total %.2f" % tuple(1e3 * v for v in (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5, t7-t6, t8-t7, t8-t0)))
