import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/python-world_amd')
import numpy as np
from oracle import resynth
from world import _hip
from world.synthesis import synthesis_device, time_axis_params
g = dict(np.load('/root/repo/tests/golden/golden_syn16k.npz'))
dat = {"f0": g["d4c_f0_after"].copy(), "vuv": g["dio_vuv"].copy(), "temporal_positions": g["tp"].copy(),
       "spectrogram": g["ct_spectrogram"].copy(), "aperiodicity": g["d4c_aperiodicity"].copy(), "fs": int(g["fs"])}
rt = _hip.Runtime.get()
tp = dat["temporal_positions"]
ny, t0, dt = time_axis_params(tp, dat["fs"])
batch = rt.make_batch([0, 0], [0, len(tp)])
args = (rt.to_device(tp), rt.to_device(dat["f0"]), rt.to_device(dat["vuv"]),
        rt.to_device(np.ascontiguousarray(dat["spectrogram"].T)),
        rt.to_device(np.ascontiguousarray(dat["aperiodicity"].T)), dat["fs"], 1024, [ny], [t0], [dt])
per = resynth.synthesis_np(dat["f0"], dat["vuv"], tp, dat["spectrogram"], dat["aperiodicity"], dat["fs"], noise=np.zeros(4 * ny))
for s in range(1, 6):
    y, _ = synthesis_device(rt, batch, *args, seed=s)
    y = y.cpu().numpy()
    np.random.seed(s)
    ref = resynth.synthesis_np(dat["f0"], dat["vuv"], tp, dat["spectrogram"], dat["aperiodicity"], dat["fs"])
    print(s, 'dev', np.mean((y - per) ** 2), 'ref', np.mean((ref - per) ** 2))
