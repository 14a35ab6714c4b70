"""The feature-head and SWIPE' blocks of bench.py on their own (config-2 batch, resident): per-head ms / GB/s / TFLOP/s and
the per-kernel split of f0_method='swipe'.  python tools/feat_bench.py   (WH_LIB=... compares library variants)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
import torch

import bench
from world.batch import WorldBatchLanes

xs = bench.make_inputs(0, 64, 16000, 10.0)
wl = WorldBatchLanes(0, lanes=1)
wl.upload(xs, 16000)
f = bench.feature_heads_block(torch, wl, 16000)
print({k: (round(v["ms"], 3), round(v.get("TFLOPs", 0), 2), round(v["GBps"])) for k, v in f.items() if isinstance(v, dict)})
s = bench.swipe_block(torch, wl, 16000)
print("swipe ms", round(s["ms"], 3), "matmul TFLOPs", round(s["matmul_TFLOPs"], 2), s["kernel_ms"])
