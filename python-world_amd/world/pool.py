"""One process, several GPUs: a host thread per device drives its own ``WorldBatch`` ("one ``ctx`` per
device per host thread; ctypes releases the GIL during calls so 8 host threads can drive 8 GPUs"; §8(e): "one host
thread + one HIP stream per device").

The reference's only fan-out is one host process pool inside Harvest (world/harvest.py:140-141) behind a single call of
``World.encode`` (world/main.py:106).  Here the batch is cut into contiguous utterance ranges balanced by samples
(``distributed.shard_ranges`` — the rule the one-process-per-GPU path shards by), one range per entry of ``devices``;
every range runs the whole pipeline on its device from its own persistent host thread — a library context, a workspace
arena and a HIP stream of its own — with NO exchange between the ranges (utterances are independent, every stage
numbers its work per utterance, the Philox stream of an utterance is keyed by its index in the whole batch), so the
results are the single-batch results bit for bit.  No ``torch.distributed``, no collectives: the per-utterance results
come back as host arrays to the one caller.

A device may appear more than once in ``devices`` (``[0, 0]``: two contexts, two host threads, two streams on one GPU)
— that is how the thread-per-device model is exercised on a one-GPU box (tests/test_hip_pool.py).
"""
import queue
import threading
import time

import numpy as np

from . import _hip
from .distributed import shard_ranges

POOL_LANE = 4000  # lanes 4000 + slot: contexts no other class hands out (their time-base contexts: 5000 + slot)


class _Worker(threading.Thread):
    """The host thread of one slot: runs the jobs it is handed, in order, with its device current."""

    def __init__(self, slot, device):
        super().__init__(name="wh-pool-%d-dev%d" % (slot, device), daemon=True)
        self.slot, self.device = slot, device
        self.jobs = queue.Queue()
        self.wb = None
        self.start()

    def run(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            fn, box, done = job
            try:
                if self.wb is None:
                    from .batch import WorldBatch
                    self.wb = WorldBatch(self.device, lane=POOL_LANE + self.slot)
                self.wb.rt.torch.cuda.set_device(self.wb.rt.device)  # (per host thread)
                box["result"] = fn(self.wb)
            except BaseException as e:  # handed to the caller of WorldBatchPool.*
                box["error"] = e
                if self.wb is not None:  # nothing of an abandoned job is left standing in this slot's context
                    try:
                        self.wb.rt.own_stream.synchronize()
                        self.wb.rt.take_flags()
                    except Exception:
                        pass
            finally:
                done.set()

    def submit(self, fn):
        box, done = {}, threading.Event()
        self.jobs.put((fn, box, done))
        return box, done


class PooledEncoding:
    """What ``WorldBatchPool.encode`` returns: the batch's contiguous utterance ranges with the resident encoding of
    each (``world.batch.BatchEncoding`` on that range's device; None for an empty range)."""

    def __init__(self, ranges, encs, n_utt):
        self.ranges, self.encs, self.n_utt = list(ranges), list(encs), int(n_utt)

    def to_dicts(self, want_ps=False, lazy=True):
        """Per-utterance dicts in batch order (``BatchEncoding.to_dicts``)."""
        out = []
        for enc in self.encs:
            if enc is not None:
                out.extend(enc.to_dicts(want_ps=want_ps, lazy=lazy))
        return out


class WorldBatchPool:
    """``devices``: the device index of every slot (default: every visible GPU once).  One persistent host thread per
    slot; ``encode`` / ``decode`` return when every slot has finished and raise the first error any slot met (after all
    of them have settled, so no work is left in flight)."""

    _shared = {}
    _shared_lock = threading.Lock()

    def __init__(self, devices=None):
        if devices is None:
            import torch
            if not torch.cuda.is_available():
                raise _hip.WorldHipError("no AMD GPU visible to PyTorch: the WORLD HIP path has no CPU fallback")
            devices = list(range(torch.cuda.device_count()))
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("WorldBatchPool needs at least one device")
        _hip.load_library()  # (fails here, on the caller's thread, when the library is missing)
        self.workers = [_Worker(s, d) for s, d in enumerate(self.devices)]
        self.timeline = []  # per slot of the last call: dict(slot, device, host_start, host_end, ev_start, ev_end)

    @classmethod
    def shared(cls, devices):
        """The process-wide pool for this ``devices`` list (threads and contexts are kept between calls)."""
        key = tuple(int(d) for d in devices)
        with cls._shared_lock:
            pool = cls._shared.get(key)
            if pool is None:
                pool = cls._shared[key] = cls(key)
            return pool

    def close(self):
        for w in self.workers:
            w.jobs.put(None)
        for w in self.workers:
            w.join(timeout=10)

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _run(self, jobs):
        """jobs: {slot: fn(wb)}.  Every job is timed on its own stream (HIP events) and on the host clock."""
        start_gate = threading.Barrier(len(jobs)) if len(jobs) > 1 else None

        def timed(fn, slot):
            def job(wb):
                torch = wb.rt.torch
                if start_gate is not None:
                    try:
                        start_gate.wait(timeout=60)  # the slots enqueue side by side, not one after the other
                    except threading.BrokenBarrierError:
                        pass  # (another slot failed before it got here: carry on alone)
                rec = {"slot": slot, "device": wb.rt.index, "host_start": time.perf_counter()}
                with wb.rt.on_stream():
                    rec["ev_start"] = torch.cuda.Event(enable_timing=True)
                    rec["ev_start"].record(torch.cuda.current_stream(wb.rt.device))
                try:
                    return fn(wb), rec
                finally:
                    with wb.rt.on_stream():
                        rec["ev_end"] = torch.cuda.Event(enable_timing=True)
                        rec["ev_end"].record(torch.cuda.current_stream(wb.rt.device))
                    wb.rt.own_stream.synchronize()
                    rec["host_end"] = time.perf_counter()
            return job

        pend = {s: self.workers[s].submit(timed(fn, s)) for s, fn in jobs.items()}
        results, err, self.timeline = {}, None, []
        for s, (box, done) in pend.items():
            done.wait()
            if "error" in box:
                err = err or box["error"]
                continue
            results[s], rec = box["result"]
            self.timeline.append(rec)
        if err is not None:
            raise err
        return results

    def overlapped(self, a=0, b=1):
        """Did the device work of slots ``a`` and ``b`` in the last call overlap in time?  On one device: by the HIP
        events recorded on their two streams (a started before b ended and b before a ended); on two devices (events of
        different devices cannot be compared): by the host clock around enqueue-to-drained."""
        ra = next(r for r in self.timeline if r["slot"] == a)
        rb = next(r for r in self.timeline if r["slot"] == b)
        if ra["device"] == rb["device"]:
            return ra["ev_start"].elapsed_time(rb["ev_end"]) > 0 and rb["ev_start"].elapsed_time(ra["ev_end"]) > 0
        return ra["host_start"] < rb["host_end"] and rb["host_start"] < ra["host_end"]

    # ---- the pipeline -------------------------------------------------------------------------------------------------
    def ranges(self, lengths):
        return shard_ranges([int(n) for n in lengths], len(self.devices))

    def encode(self, xs, fs, **kw):
        """``WorldBatch.encode`` of the utterance list ``xs`` cut over the slots -> ``PooledEncoding``."""
        ranges = self.ranges([len(x) for x in xs])
        kw = dict(kw)
        kw.pop("check", None)

        def job(a, b):
            def run(wb):
                enc = wb.encode(list(xs[a:b]), fs, check=False, **kw)
                wb.rt.own_stream.synchronize()
                return wb.settle_encode(enc, "WorldBatchPool.encode")
            return run

        res = self._run({s: job(a, b) for s, (a, b) in enumerate(ranges) if b > a})
        return PooledEncoding(ranges, [res.get(s) for s in range(len(ranges))], len(xs))

    def decode(self, penc, seed=0, noise=None, **kw):
        """``WorldBatch.decode_device`` of every range of a ``PooledEncoding`` -> list of per-utterance host waveforms
        (own pageable arrays), in batch order.  The device noise of utterance u is the one the single batch draws for it
        (``synthesis.philox_seed_for_offset``); Requiem ranges chain the noise cursor through the ranges like
        consecutive reference calls would (``cursor=``: where the first utterance starts; default 0)."""
        plan, tps = [], []
        for s, ((a, b), enc) in enumerate(zip(penc.ranges, penc.encs)):
            if enc is None:
                continue
            plan.append((s, a, b, (lambda wb, enc=enc: enc)))
            fo, tp_h = enc.batch.frame_off, enc.host_times()
            tps.append([tp_h[int(fo[u]):int(fo[u + 1])] for u in range(enc.n_utt)])
        first = next(e for e in penc.encs if e is not None)
        return self._decode(plan, tps, first.fs, first.is_requiem, seed, noise, kw)

    def decode_dicts(self, dats, seed=0, noise=None, **kw):
        """The same for a list of encode() dicts (reference layout or ``EncodingDict``s): every slot uploads — or, for
        dense values that never left its device, slices — its range itself (``BatchEncoding.from_dicts``).  Dicts that
        came out of ``encode`` on this pool go back to the slots that hold them."""
        from .batch import BatchEncoding, EncodingDict

        n = len(dats)
        runs, ok = [], True  # runs of dicts that share a resident encoding of one of this pool's slots
        for i, d in enumerate(dats):
            enc = getattr(d, "_enc", None) if isinstance(d, EncodingDict) else None
            slot = None
            if enc is not None:
                s = enc.rt.lane - POOL_LANE
                if 0 <= s < len(self.workers) and self.devices[s] == enc.rt.index:
                    slot = s
            if slot is None:
                ok = False
                break
            if runs and runs[-1][0] == slot and runs[-1][3] is enc:
                runs[-1][2] = i + 1
            else:
                runs.append([slot, i, i + 1, enc])
        if ok and len({r[0] for r in runs}) == len(runs):
            parts = [(r[0], r[1], r[2]) for r in runs]
        else:
            parts = [(s, a, b) for s, (a, b) in enumerate(shard_ranges([len(d['f0']) for d in dats], len(self.workers)))
                     if b > a]
        plan = [(s, a, b, (lambda wb, a=a, b=b: BatchEncoding.from_dicts(wb.rt, dats[a:b]))) for s, a, b in parts]
        tps = [[np.asarray(d['temporal_positions'], dtype=np.float64) for d in dats[a:b]] for _, a, b in parts]
        assert sum(b - a for _, a, b in parts) == n
        return self._decode(plan, tps, dats[0]['fs'], bool(dats[0]['is_requiem']), seed, noise, kw)

    def _decode(self, plan, tps, fs, is_requiem, seed, noise, kw):
        from .synthesis import philox_seed_for_offset

        kw = dict(kw)
        kw.pop("check", None)
        cursors = None
        if is_requiem:
            # the utterances consume the circular noise seed one after the other (world/synthesisRequiem.py:131-141): a
            # range starts where the ranges before it stop — known on the host from the output lengths, before any decode
            from .synthesisRequiem import cursor_after, seed_table_shape
            nlen, nb = seed_table_shape(fs, kw.get("seeds"))
            cur = kw.pop("cursor", None)
            cur = np.zeros(nb) if cur is None else np.array(cur, dtype=np.float64)
            cursors = []
            for part in tps:
                cursors.append(cur)
                cur = cursor_after(part, fs, cur, nlen)

        def job(k, a, b, get_enc):
            def run(wb):
                enc = get_enc(wb)
                kw_s = dict(kw, seed=philox_seed_for_offset(seed, a))
                if noise is not None:
                    kw_s["noise"] = noise[a:b]
                if cursors is not None:
                    kw_s["cursor"] = cursors[k]
                y, y_off = wb.decode_device(enc, check=True, **kw_s)
                with wb.rt.on_stream():
                    host = wb.rt.to_host(y)
                return [np.array(host[int(y_off[u]):int(y_off[u + 1])]) for u in range(len(y_off) - 1)]
            return run

        res = self._run({s: job(k, a, b, g) for k, (s, a, b, g) in enumerate(plan)})
        out = []
        for s, a, b, _ in plan:
            out.extend(res[s])
        return out
