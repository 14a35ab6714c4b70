"""GPU: one process, a host thread per device (world.pool.WorldBatchPool).  A one-GPU box
exercises it with ``devices=[0, 0]``: two library contexts, two host threads, two HIP streams on the one device, the
C-ABI entered from both threads at the same time.  Checked: the results are the single-batch results BIT FOR BIT
(encode tensors and decoded samples, pulse-wise and Requiem), the two threads' device work overlapped in time (HIP
events on the two streams), and a host thread makes progress while another one sits inside a blocking C-ABI call (ctypes
releases the GIL).  The reference's counterpart is one call in one process, world/main.py:106, whose only fan-out is
the process pool of world/harvest.py:140-141."""
import threading
import time

import numpy as np
import pytest

from conftest import synth_cached

pytestmark = pytest.mark.gpu

DENSE = ('temporal_positions', 'f0', 'vuv', 'spectrogram', 'aperiodicity')


def _batch(n, fs=16000, base=700):
    return [synth_cached(base + i, fs, 1.0 + 0.25 * (i % 5)) for i in range(n)]


@pytest.mark.parametrize("method,requiem", [("dio", False), ("harvest", True)])
def test_two_host_threads_on_one_device_equal_the_single_batch(method, requiem):
    from world.main import World
    from world.pool import WorldBatchPool

    fs = 16000
    xs = _batch(24)
    w = World()
    ref = w.encode_batch(fs, xs, f0_method=method, is_requiem=requiem)
    ref_vals = [{k: np.array(d[k]) for k in DENSE} for d in ref]
    w.decode_batch(ref, seed=5)
    pool = WorldBatchPool.shared([0, 0])
    overlapped = False
    for _ in range(3):  # (the first call also builds tables and arenas; overlap is a property of the steady state)
        got = w.encode_batch(fs, xs, f0_method=method, is_requiem=requiem, devices=[0, 0])
        overlapped = overlapped or pool.overlapped(0, 1)
    assert len(got) == len(xs)
    assert overlapped, "the two host threads' kernels did not overlap on the device"
    assert [r["slot"] for r in pool.timeline] == [0, 1] and len({id(wk.wb.rt.ctx) for wk in pool.workers}) == 2
    assert pool.workers[0].wb.rt.own_stream.cuda_stream != pool.workers[1].wb.rt.own_stream.cuda_stream
    # decode first: the dense values have not left the device, every slot decodes what it holds
    w.decode_batch(got, seed=5, devices=[0, 0])
    for u, (d, r) in enumerate(zip(got, ref)):
        assert d['out'].flags.owndata and np.array_equal(d['out'], r['out']), u
    for u, (d, r) in enumerate(zip(got, ref_vals)):
        for k in DENSE:
            assert np.array_equal(d[k], r[k]), (k, u)
    # and from caller-built dicts (everything uploaded again, ranges balanced by frames): the same samples
    plain = [dict(d.items()) for d in got]
    for d in plain:
        d.pop('out')
    w.decode_batch(plain, seed=5, devices=[0, 0])
    for u, (d, r) in enumerate(zip(plain, ref)):
        assert np.array_equal(d['out'], r['out']), u
    for wk in pool.workers:
        assert wk.wb.rt.take_flags() == [0] * 16


def test_three_slots_and_an_empty_range():
    """More slots than utterances: an empty range takes no part; the rest equals the single batch."""
    from world.batch import WorldBatch
    from world.pool import WorldBatchPool

    fs = 16000
    xs = _batch(2, base=760)
    enc = WorldBatch().encode(xs, fs, f0_method='dio')
    pool = WorldBatchPool([0, 0, 0])
    try:
        penc = pool.encode(xs, fs, f0_method='dio')
        sizes = [b - a for a, b in penc.ranges]
        assert sorted(sizes) == [0, 1, 1] and penc.encs[sizes.index(0)] is None
        dicts, ref = penc.to_dicts(lazy=False), enc.to_dicts()
        for d, r in zip(dicts, ref):
            for k in DENSE:
                assert np.array_equal(d[k], r[k]), k
        ys = pool.decode(penc, seed=3)
        y_ref, off = WorldBatch().decode_device(enc, seed=3)
        y_ref = y_ref.cpu().numpy()
        for u, y in enumerate(ys):
            assert np.array_equal(y, y_ref[off[u]:off[u + 1]])
    finally:
        pool.close()


def test_an_error_in_one_slot_reaches_the_caller_and_leaves_nothing_behind():
    from world._hip import WorldHipError
    from world.pool import WorldBatchPool

    fs = 16000
    pool = WorldBatchPool.shared([0, 0])
    xs = _batch(4, base=780)
    xs[3] = np.zeros(8)  # shorter than any stage accepts: the library refuses the second slot's range
    with pytest.raises(WorldHipError):
        pool.encode(xs, fs, f0_method='harvest')
    good = pool.encode(_batch(4, base=780), fs, f0_method='harvest')  # both slots are usable afterwards
    assert all(e is not None for e in good.encs)
    for wk in pool.workers:
        assert wk.wb.rt.take_flags() == [0] * 16


def test_the_gil_is_released_inside_blocking_cabi_calls():
    """Thread A enqueues ~100 ms of device work and waits for it inside ONE C-ABI call (wh_take_flags: a stream
    synchronise); thread B — plain Python — keeps stamping the clock.  If ctypes held the GIL across the call, B would
    stand still for its whole length."""
    from world.batch import WorldBatch

    fs = 16000
    xs = [synth_cached(i, fs, 10.0) for i in range(64)]  # (the utterances the full-size tests use: generated once)
    wb = WorldBatch(lane=7)
    batch, x_d, tp_d = wb.upload(xs, fs)
    wb.encode_device(batch, x_d, tp_d, fs, f0_method='harvest', check=True)  # warm: tables, arena
    stamps, stop = [], threading.Event()

    def ticker():
        while not stop.is_set():
            stamps.append(time.perf_counter())
            time.sleep(0.0002)

    th = threading.Thread(target=ticker)
    for _ in range(6):
        wb.encode_device(batch, x_d, tp_d, fs, f0_method='harvest', check=False)
    th.start()
    try:
        t0 = time.perf_counter()
        wb.rt.take_flags()  # blocks in the library until the six encodes (~10 ms each) have drained
        t1 = time.perf_counter()
    finally:
        stop.set()
        th.join()
    blocked = t1 - t0
    inside = np.array([s for s in stamps if t0 <= s <= t1])
    assert blocked > 0.02, "the call did not block long enough to tell (%.1f ms)" % (blocked * 1e3)
    assert len(inside) >= 20, "the other host thread made no progress during the blocking call"
    assert np.max(np.diff(np.r_[t0, inside, t1])) < 0.5 * blocked
