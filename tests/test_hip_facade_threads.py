"""GPU: the drop-in facade called from several Python threads at once.  The reference is plain NumPy — two threads may sit in
World.encode together; here the drop-ins share the process's default library context (one arena, one stream) and ctypes
releases the GIL inside every call, so they serialise on world._hip.FACADE_LOCK: each thread gets, bit for bit, what it gets
alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_facade_calls_equal_the_serial_results():
    from world._synthetic import synth_utterance
    from world.main import World

    fs = 16000
    xs = [synth_utterance(400 + i, fs, 0.5 + 0.1 * i) for i in range(4)]
    kinds = [("harvest", True), ("dio", False), ("harvest", False), ("dio", True)]
    w = World()
    keys = ("temporal_positions", "f0", "vuv", "spectrogram", "aperiodicity")
    serial = [w.encode(fs, x, f0_method=m, is_requiem=r) for x, (m, r) in zip(xs, kinds)]
    serial_batch = w.encode_batch(fs, xs, f0_method="dio")
    got, errs = [[] for _ in xs], []

    def work(i):
        try:
            for rep in range(6):
                d = World().encode(fs, xs[i], f0_method=kinds[i][0], is_requiem=kinds[i][1])
                got[i].append({k: np.array(d[k]) for k in keys})
                if rep % 3 == 0:  # a batched call in between, from this thread too
                    b = World().encode_batch(fs, xs, f0_method="dio")
                    got[i].append(("batch", [{k: np.array(e[k]) for k in keys} for e in b]))
        except BaseException as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    assert all(not t.is_alive() for t in ts)
    for i in range(len(xs)):
        assert len(got[i]) == 8
        for item in got[i]:
            if isinstance(item, tuple):
                for e, ref in zip(item[1], serial_batch):
                    for k in keys:
                        assert np.array_equal(e[k], ref[k]), (i, "batch", k)
            else:
                for k in keys:
                    assert np.array_equal(item[k], serial[i][k]), (i, k)
