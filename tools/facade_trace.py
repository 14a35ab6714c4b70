#!/usr/bin/env python3
"""Host-side timeline of the drop-in face's resynthesis flow (World.encode_batch -> scale -> decode_batch, 64 x 10 s):
when each host call starts and returns (no synchronisation added), for the two-part and the single-batch form.
python tools/facade_trace.py"""
import sys, time, functools
sys.path.insert(0, "."); sys.path.insert(0, "python-world_amd")
import torch
import bench
from world import main, batch, _hip

T0 = [0.0]
LOG = []


def traced(owner, name, label=None):
    fn = getattr(owner, name)

    @functools.wraps(fn)
    def w(*a, **kw):
        t = time.perf_counter()
        r = fn(*a, **kw)
        LOG.append((label or name, (t - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3))
        return r
    setattr(owner, name, w)


traced(batch.WorldBatch, "upload")
traced(batch.WorldBatch, "encode_device")
traced(batch.BatchEncoding, "to_dicts")
traced(batch.WorldBatch, "decode_device")
traced(batch.WorldBatch, "settle_decode")
traced(batch.WorldBatchPipeline, "synchronize", "pipe.synchronize")
traced(_hip.Runtime, "to_host")
traced(_hip.Runtime, "to_device_concat")
traced(_hip.Runtime, "make_batch")
traced(_hip.Runtime, "to_device")
_fd = batch.BatchEncoding.from_dicts.__func__
def fd(cls, rt, dats):
    t = time.perf_counter()
    r = _fd(cls, rt, dats)
    LOG.append(("from_dicts", (t - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3))
    return r
batch.BatchEncoding.from_dicts = classmethod(fd)

xs = bench.make_inputs(0, 64, 16000, 10.0)
W = main.World()


def flow():
    dats = W.encode_batch(16000, xs, f0_method="dio")
    LOG.append(("encode_batch returns", (time.perf_counter() - T0[0]) * 1e3, 0))
    for d in dats:
        W.scale_pitch(d, 1.5)
        W.scale_duration(d, 2.0)
    LOG.append(("scaled", (time.perf_counter() - T0[0]) * 1e3, 0))
    return W.decode_batch(dats)


FORMS = (("two parts", 16 << 20), ("single batch", 1 << 62))
if len(sys.argv) > 1:  # "parts" / "single": that form only (under rocprofv3: the trace's last flow is unambiguous)
    FORMS = [f for f in FORMS if f[0].startswith({"parts": "two", "single": "single"}[sys.argv[1]])]
for label, split in FORMS:
    main.FACADE_SPLIT_BYTES = split
    for rep in range(4):
        torch.cuda.synchronize()
        del LOG[:]
        T0[0] = time.perf_counter()
        flow()
        torch.cuda.synchronize()
        total = (time.perf_counter() - T0[0]) * 1e3
    print("== %s: %.2f ms" % (label, total))
    for name, a, b in LOG:
        print("  %-22s %7.2f -> %7.2f" % (name, a, b) if b else "  %-22s %7.2f" % (name, a))
