"""Race / memory-safety evidence from the product's own outputs (the sanitizer pass this image allows is
tools/asan_probe.sh: profiles/r05_asan_*.log).  Checks over the whole pipeline:

* determinism: every output is BITWISE identical from run to run (a data race in an LDS exchange, a ballot compaction or
  an XCD-remapped grid shows up as run-to-run differences) — since round 5 including the decode, whose overlap-add is
  summed in a fixed order (rows of runs + gather) instead of by FP64 atomics — and an utterance decodes to the same
  samples alone, anywhere in a batch and on a shard of it;
* guard bands: the caller-visible output buffers are allocated with NaN-patterned guard regions on both sides and
  the guards are intact after the kernels ran (out-of-bounds stores past either end of an output);
* poisoned scratch: outputs do not depend on what the workspace arena and the outputs held before the call
  (reads of uninitialised scratch / stale results of an earlier call)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 16000


def _batch(n=6):
    from world._synthetic import synth_utterance

    return [synth_utterance(300 + i, FS, 0.6 + 0.17 * i) for i in range(n)]


@pytest.mark.parametrize("method,requiem", [("dio", False), ("harvest", True), ("swipe", False)])
def test_encode_is_bitwise_deterministic(method, requiem):
    from world.batch import WorldBatch

    xs = _batch()
    wb = WorldBatch()
    runs = []
    for r in range(4):
        enc = wb.encode(xs, FS, f0_method=method, is_requiem=requiem)
        runs.append([t.cpu().numpy().copy() for t in (enc.f0, enc.vuv, enc.spectrogram, enc.aperiodicity)])
        if r == 1:  # perturb the allocator / scratch state between runs
            wb.encode(xs[:2], FS, f0_method="dio")
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("requiem", [False, True])
def test_decode_is_bitwise_deterministic(requiem):
    """The overlap-add of the pulse responses / Requiem frames is summed in a fixed order (rows of runs of pulses /
    frames, then a gather in run order: wh_synthesis.hip RunState, req_filter_kernel) — no atomics since round 5, so the
    decode repeats bit for bit (rounds 1-4: equal up to the order of the FP64 atomics, 1e-17)."""
    from world.batch import WorldBatch

    xs = _batch(4)
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="dio", is_requiem=requiem)
    outs = []
    for r in range(4):
        y, y_off = wb.decode_device(enc, seed=7)
        outs.append(y.cpu().numpy().copy())
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    # host-noise path (reference-parity mode): the same property with caller-supplied noise
    if not requiem:
        rng = np.random.RandomState(3)
        noise = [rng.randn(2 * len(x)) for x in xs]
        a = wb.decode_device(enc, noise=noise)[0].cpu().numpy()
        b = wb.decode_device(enc, noise=noise)[0].cpu().numpy()
        assert np.array_equal(a, b)
        # a different Philox seed really changes the audio (the determinism above is not a constant output)
        c = wb.decode_device(enc, seed=8)[0].cpu().numpy()
        assert np.max(np.abs(c - outs[0])) > 1e-6


@pytest.mark.parametrize("requiem", [False, True])
def test_decode_of_an_utterance_does_not_depend_on_its_batch(requiem):
    """SURVEY §7.1c "sharded == unsharded bitwise", now for the decode too: the overlap-add runs are numbered per
    utterance, so an utterance decoded alone, in the middle of a batch or on another rank's shard gives the same samples
    bit for bit (host-supplied noise / explicit Requiem cursors, so that the random streams are the same)."""
    from world.batch import WorldBatch
    from world.synthesisRequiem import _advance, _default_seeds

    xs = _batch(5)
    wb = WorldBatch()
    enc = wb.encode(xs, FS, f0_method="dio", is_requiem=requiem)
    rng = np.random.RandomState(11)
    noise = [rng.randn(2 * len(x)) for x in xs]
    kw = {} if requiem else {"noise": noise}
    y, y_off = wb.decode_device(enc, **kw)
    y = y.cpu().numpy()
    cur = np.zeros(3)
    for u, x in enumerate(xs):
        alone = WorldBatch()
        e1 = alone.encode([x], FS, f0_method="dio", is_requiem=requiem)
        k1 = {"cursor": cur} if requiem else {"noise": [noise[u]]}
        y1 = alone.decode_device(e1, **k1)[0].cpu().numpy()
        assert np.array_equal(y1, y[int(y_off[u]):int(y_off[u + 1])]), u
        if requiem:
            nlen = int(_default_seeds[(FS, wb.rt.index, wb.rt.lane)]["noise_d"].shape[0])
            cur = _advance(cur, int(y_off[u + 1] - y_off[u]), nlen)
    # two "ranks": the batch split into shards decodes to the same samples as the whole
    for lo, hi in ((0, 2), (2, 5)):
        es = WorldBatch().encode(xs[lo:hi], FS, f0_method="dio", is_requiem=requiem)
        if requiem:
            c0 = np.zeros(3)
            for u in range(lo):
                c0 = _advance(c0, int(y_off[u + 1] - y_off[u]), nlen)
            ys = wb.decode_device(es, cursor=c0)[0].cpu().numpy()
        else:
            ys = wb.decode_device(es, noise=noise[lo:hi])[0].cpu().numpy()
        assert np.array_equal(ys, y[int(y_off[lo]):int(y_off[hi])]), (lo, hi)


def test_guard_bands_around_outputs_stay_intact():
    """Stage entry points write into caller buffers: carve them out of a NaN-filled slab and check the slab outside."""
    import ctypes

    from world import _hip, _tables
    from world.cheaptrick import default_fft_size

    rt = _hip.Runtime.get()
    torch = rt.torch
    xs = _batch(3)
    lens = [len(x) for x in xs]
    nfs = [_tables.frame_count(n, FS, 5) for n in lens]
    batch = rt.make_batch(np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(nfs)]))
    x_d = rt.to_device(np.concatenate(xs))
    tp_d = rt.to_device(np.concatenate([_tables.frame_times(n, 5) for n in nfs]))
    nf = batch.total_frames
    fft = default_fft_size(FS)
    k = fft // 2 + 1
    guard = 4096
    sentinel = float(np.frombuffer(np.uint64(0x7FF8DEADBEEF0001).tobytes(), dtype=np.float64)[0])

    def slab(n):
        s = torch.full((n + 2 * guard,), float("nan"), dtype=torch.float64, device=rt.device)
        s.view(torch.int64)[:] = 0x7FF8DEADBEEF0001
        return s, s[guard:guard + n]

    def intact(s, n):
        raw = s.view(torch.int64)
        return bool((raw[:guard] == 0x7FF8DEADBEEF0001).all()) and bool((raw[guard + n:] == 0x7FF8DEADBEEF0001).all())

    from world.dio import dio_device
    from world.stonemask import stonemask_device

    f0_d, vuv_d, _, _ = dio_device(rt, batch, x_d, tp_d, FS, 71, 800, 2, 4000, 5, 0.1)
    f0_d = stonemask_device(rt, batch, x_d, tp_d, f0_d, FS, 71)
    s_spec, spec = slab(nf * k)
    s_ap, ap = slab(nf * k)
    _hip.check(rt.lib.wh_cheaptrick(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d),
                                    rt.ptr(vuv_d), float(FS), fft, -0.15, rt.ptr(spec), ctypes.c_void_p(None)))
    _hip.check(rt.lib.wh_d4c(rt.ctx, rt.stream(), batch.handle, rt.ptr(x_d), rt.ptr(tp_d), rt.ptr(f0_d), rt.ptr(vuv_d),
                             float(FS), 0.85, fft, rt.ptr(ap), ctypes.c_void_p(None)))
    torch.cuda.synchronize()
    assert intact(s_spec, nf * k) and intact(s_ap, nf * k)
    assert bool(torch.isfinite(spec).all()) and bool(torch.isfinite(ap).all())  # every element inside was written
    # synthesis output
    from world.synthesis import synthesis_device, time_axis_params

    tp_h = tp_d.cpu().numpy()
    fo = batch.frame_off
    geo = [time_axis_params(tp_h[int(fo[u]):int(fo[u + 1])], FS) for u in range(batch.n_utt)]
    ny = [g[0] for g in geo]
    y, y_off = synthesis_device(rt, batch, tp_d, f0_d, vuv_d, spec.view(nf, k), ap.view(nf, k), FS, fft, ny,
                                [g[1] for g in geo], [g[2] for g in geo], seed=5)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    assert rt.take_flags() == [0] * 16
    del sentinel


def test_outputs_do_not_depend_on_stale_scratch():
    """Run a LARGE unrelated batch (fills the workspace arena and torch's cached blocks with other data), then the
    small batch again: results must equal the first run bit for bit."""
    from world.batch import WorldBatch
    from world._synthetic import synth_utterance

    wb = WorldBatch()
    xs = _batch(3)
    first = wb.encode(xs, FS, f0_method="harvest")
    ref = [t.cpu().numpy().copy() for t in (first.f0, first.vuv, first.spectrogram, first.aperiodicity)]
    del first
    big = [synth_utterance(400 + i, FS, 2.0) for i in range(12)]
    e2 = wb.encode(big, FS, f0_method="harvest", is_requiem=True)
    wb.decode_device(e2)
    del e2
    again = wb.encode(xs, FS, f0_method="harvest")
    for a, t in zip(ref, (again.f0, again.vuv, again.spectrogram, again.aperiodicity)):
        assert np.array_equal(a, t.cpu().numpy())
