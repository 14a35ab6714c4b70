import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "python-world_amd"))
import torch, bench
from world.batch import WorldBatchLanes
xs = bench.make_inputs(0, 64, 16000, 10.0)
wl = WorldBatchLanes(0, lanes=1); wl.upload(xs, 16000)
f = bench.feature_heads_block(torch, wl, 16000)
print({k: (round(v["ms"], 3), round(v.get("TFLOPs", 0), 2), round(v["GBps"])) for k, v in f.items() if isinstance(v, dict)})
s = bench.swipe_block(torch, wl, 16000)
print("swipe ms", round(s["ms"], 3), "matmul TFLOPs", round(s["matmul_TFLOPs"], 2), s["kernel_ms"])
