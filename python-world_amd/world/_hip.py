"""ctypes binding of libworld_hip.so (the C-ABI declared in include/world_hip.h) plus the small
amount of device plumbing the stage shims need.  PyTorch is used only for device memory and
streams (torch.cuda tensors are handed to the library as raw device pointers).

There is deliberately NO CPU fallback: if the library or a GPU is missing every stage raises.
"""
import contextlib
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WH_LIB selects another build of the same library (tools/build_variants.py: kernel tuning variants)
LIB_PATH = os.environ.get("WH_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libworld_hip.so")

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_dbl = ctypes.c_double
_int = ctypes.c_int

# name -> (restype, argtypes); every symbol include/world_hip.h declares
SIGNATURES = {
    "wh_version": (_int, []),
    "wh_last_error": (ctypes.c_char_p, []),
    "wh_device_count": (_int, [ctypes.POINTER(_int)]),
    "wh_ctx_create": (_int, [_int, ctypes.POINTER(_vp)]),
    "wh_ctx_destroy": (_int, [_vp]),
    "wh_ctx_trim": (_int, [_vp]),
    "wh_malloc": (_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    "wh_free": (_int, [_vp]),
    "wh_memcpy_h2d": (_int, [_vp, _vp, ctypes.c_size_t, _vp]),
    "wh_memcpy_d2h": (_int, [_vp, _vp, ctypes.c_size_t, _vp]),
    "wh_memset": (_int, [_vp, _int, ctypes.c_size_t, _vp]),
    "wh_stream_sync": (_int, [_vp]),
    "wh_host_alloc": (_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    "wh_host_free": (_int, [_vp]),
    "wh_copy_mapped": (_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, _int]),
    "wh_batch_create": (_int, [_vp, _int, _c_i64p, _c_i64p, ctypes.POINTER(_vp)]),
    "wh_batch_destroy": (_int, [_vp]),
    "wh_num_frames": (ctypes.c_int64, [ctypes.c_int64, _dbl, _dbl]),
    "wh_cheaptrick": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _dbl, _vp, _vp]),
    "wh_profile_enable": (_int, [_vp, _int]),
    "wh_profile_collect": (_int, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_float), _int,
                                  ctypes.POINTER(_int)]),
    "wh_take_flags": (_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int32)]),
    "wh_bounds_build": (_int, []),
    "wh_bounds_last": (_int, [_c_i64p]),
    "wh_bounds_selftest": (_int, [_vp, _vp]),
    "wh_math_probe": (_int, [_vp, _vp, _int, _vp, _vp, ctypes.c_int64]),
    "wh_flags_post": (_int, [_vp, _vp, _int]),
    "wh_flags_poll": (_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "wh_dio": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _int,
                      _vp, _vp, _vp, _vp]),
    "wh_harvest": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp]),
    "wh_harvest_set_event_caps": (_int, [_vp, _vp, ctypes.c_int64]),
    "wh_harvest_event_counts": (_int, [_vp, _vp, _vp, ctypes.c_int64]),
    "wh_stonemask": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _int, _vp]),
    "wh_synthesis": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp,
                            ctypes.c_uint64, _vp, _vp]),
    "wh_synthesis_timebase": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, ctypes.c_int64, _dbl]),
    "wh_synthesis_render": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp,
                                   ctypes.c_uint64, _vp, _vp]),
    "wh_philox_normals": (_int, [_vp, _vp, ctypes.c_uint64, _int, ctypes.c_int64, ctypes.c_int64, _vp]),
    "wh_cumsum_exact": (_int, [_vp, _vp, _vp, _vp, _int]),
    "wh_peak_normalise": (_int, [_vp, _vp, _vp, _vp, _int]),
    "wh_synthesis_plan": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    "wh_synthesis_requiem": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp, _vp, ctypes.c_int64,
                                    _vp, _int, _vp, ctypes.c_int64, _int, _vp, _vp]),
    "wh_d4c": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _int, _vp, _vp]),
    "wh_d4c_bands": (_int, [_dbl, _int]),
    "wh_d4c_requiem": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _int, _vp]),
    "wh_feature_matmul": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, ctypes.c_int64, _int, _vp, _dbl, _vp, _int, _int, _vp,
                                 ctypes.c_int64]),
    "wh_feature_matmul_tagged": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, ctypes.c_int64, _int, _vp, _dbl, _vp, _int, _int,
                                        _vp, ctypes.c_int64, ctypes.c_uint64]),
    "wh_context_frames": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _int, _vp]),
    "wh_warp_spectrum": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _vp, _vp, _vp]),
    "wh_modify_duration": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "wh_pcm16_to_f64": (_int, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    "wh_f64_to_pcm16": (_int, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    "wh_swipe": (_int, [_vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _int, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp]),
    "wh_requiem_seeds": (_int, [_vp, _vp, _dbl, _int, ctypes.c_int64, _int, ctypes.c_uint64, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


class WorldHipError(RuntimeError):
    pass


# The drop-in functions (world.harvest.harvest, world.cheaptrick.cheaptrick, ..., World.encode_batch / decode_batch) run
# on the process's default context: one arena, one stream, one set of flags.  The reference is plain NumPy and may be called
# from several threads at once; two threads inside the same library context would lay its scratch out over each other
# (ctypes releases the GIL during the calls).  So the drop-ins take this lock: concurrent callers are served one after the
# other and get the results they would get alone.  (Parallelism within a process: world.pool / devices=[...] — contexts of
# their own — or one WorldBatch lane per thread.)
FACADE_LOCK = threading.RLock()


def serialised(fn):
    """Decorator of a drop-in entry point: holds FACADE_LOCK for the duration of the call (re-entrant)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        with FACADE_LOCK:
            return fn(*a, **kw)
    return wrapped


def same_frames(where, dense=(), **per_frame):
    """The per-frame arrays a stage function is handed must be 1-D and of one length, the dense ones (bins, frames) over it:
    the kernels index all of them by the batch's frame count, so a short one would be read past its end (the reference fails
    inside NumPy on such input).  Returns the frame count."""
    shapes = {k: np.shape(v) for k, v in per_frame.items()}
    lens = {s[0] if len(s) == 1 else None for s in shapes.values()}
    if len(lens) != 1 or None in lens:
        raise ValueError("%s: per-frame arrays must be 1-D and of one length, got %s" % (where, shapes))
    n = lens.pop()
    for name, v in dense:
        if np.ndim(v) != 2 or np.shape(v)[1] != n:
            raise ValueError("%s: '%s' must be (bins, %d frames), got %s" % (where, name, n, np.shape(v)))
    return n


# WH_FLAG_* of include/world_hip.h: sticky conditions raised by kernels instead of failing silently
FLAG_STONEMASK_WINDOW, FLAG_EVENT_OVERFLOW, FLAG_NOISE_SHORT, FLAG_NO_PULSE, FLAG_PULSE_OVERFLOW, FLAG_OOB = range(6)
FLAG_MESSAGES = {
    FLAG_STONEMASK_WINDOW: "StoneMask: a frame's f0 needs a longer analysis window than the one sized from min_f0 "
                           "(frame left unrefined)",
    FLAG_EVENT_OVERFLOW: "a zero-crossing list of Harvest exceeded its estimated capacity (stretches that are constant up "
                         "to rounding, e.g. digital silence next to signal): the checked calls — harvest(), "
                         "encode_device(check=True), World.encode / encode_batch — repeat themselves with the counted "
                         "capacities; a call kept asynchronous takes event_caps='safe' or world.harvest.counted_event_caps(rt)",
    FLAG_NOISE_SHORT: "synthesis ran out of host-supplied noise samples",
    FLAG_NO_PULSE: "an utterance produced no pulse (the reference asserts, world/synthesis.py:131)",
    FLAG_PULSE_OVERFLOW: "more pulses (or overlap-add rows) than pulse_cap provides for: trailing pulses / runs were dropped "
                         "(pass pulse_cap=world.synthesis.safe_pulse_cap(ny))",
    FLAG_OOB: "bounds build: a kernel indexed outside one of its buffers (world._hip.bounds_last() has the record)",
}


def bounds_build():
    """True when the loaded library is the bounds build (tools/build_variants.py ...:-DWH_BOUNDS=1)."""
    return bool(load_library().wh_bounds_build())


def bounds_last():
    """(count, buffer tag, element index, buffer size) of the out-of-range accesses the last take_flags() found."""
    buf = (ctypes.c_int64 * 4)()
    check(load_library().wh_bounds_last(buf))
    return tuple(int(v) for v in buf)


def load_library():
    """dlopen libworld_hip.so and attach the prototypes.  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise WorldHipError("libworld_hip.so is missing (%s): build it with `python python-world_amd/build.py`; "
                                "there is no CPU fallback" % LIB_PATH)
        try:
            import torch  # noqa: F401  -- make sure torch's HIP runtime is the one already mapped
        except Exception:
            pass
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def table_tag(*arrays):
    """64-bit content tag (never 0) of host tables for wh_feature_matmul_tagged / wh_swipe_window.table_tag: computed
    ONCE where a cached table is built, so that equal tags mean equal content by construction."""
    import hashlib

    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return (int.from_bytes(h.digest(), "little") >> 2) | 1  # the library derives tag*2+1 / tag*2+2 from it


def check(rc):
    if rc != 0:
        raise WorldHipError(load_library().wh_last_error().decode("utf-8", "replace"))


_pool = None


class _copy_pool:
    """The process-wide 16-thread pool for host-side staging copies (created on first use, kept)."""

    def __enter__(self):
        global _pool
        if _pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _pool = ThreadPoolExecutor(max_workers=16, thread_name_prefix="wh-stage")
        return _pool

    def __exit__(self, *exc):
        return False


class Runtime:
    """One wh_ctx per (device, lane); device buffers are torch tensors.

    Lane 0 launches on torch's current stream.  Every further lane owns a private HIP stream and a private
    context (workspace, tables, flags), so that independent sub-batches can be in flight on the GPU at the same
    time: the chip-filling kernels of one lane overlap the latency-bound per-utterance kernels (IIR chains,
    the exact phase scan, pulse compaction) of another.
    """

    _instances = {}
    _instances_lock = threading.Lock()  # (world.pool creates runtimes from several host threads)

    def __init__(self, device_index, lane=0):
        import torch

        if not torch.cuda.is_available():
            raise WorldHipError("no AMD GPU visible to PyTorch: the WORLD HIP path has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.index = device_index
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)  # force primary-context creation before our first hip call
        h = _vp()
        check(self.lib.wh_ctx_create(device_index, ctypes.byref(h)))
        self.ctx = h
        self.lane = lane
        # (a high-priority stream for the lanes whose short serial kernels are meant to run under another lane's
        # chip-filling ones was measured and is worse: config 2 10.12 against 9.83 ms with the time-base lane at -1)
        self.own_stream = torch.cuda.Stream(device=self.device) if lane else None
        self.harvest_lists = 0  # utterances x channels of this context's last wh_harvest (world.harvest.counted_event_caps)
        # a context is driven by one host thread at a time: WorldBatch methods hold this for the duration of a call, so two
        # threads that share a lane (two WorldBatch() on the default one) enqueue whole calls one after the other
        self.lock = threading.RLock()

    @classmethod
    def get(cls, device_index=None, lane=0):
        import torch

        if device_index is None:
            device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
        with cls._instances_lock:
            rt = cls._instances.get((device_index, lane))
            if rt is None:
                rt = cls(device_index, lane)
                cls._instances[(device_index, lane)] = rt
        return rt

    def trim(self):
        """Give this lane's scratch back to the device (wh_ctx_trim: the arena and per-call buffers only grow)."""
        check(self.lib.wh_ctx_trim(self.ctx))
        if hasattr(self, "timebase_generation"):  # a time-base context (world.batch): what it held is gone —
            self.timebase_generation += 1         # encodings that point at it decode with an in-line time base

    @classmethod
    def trim_all(cls):
        """trim() every runtime of the process and empty torch's cache: between phases of very different batch sizes."""
        for rt in list(cls._instances.values()):
            rt.trim()
        try:
            import torch
            torch.cuda.empty_cache()
        except Exception:
            pass

    def on_stream(self):
        """Context manager: make this lane's stream torch's current stream (no-op for lane 0)."""
        if self.own_stream is None:
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.own_stream)

    # ---- memory ---------------------------------------------------------------------------
    def stream(self):
        if self.own_stream is not None:
            return _vp(self.own_stream.cuda_stream)
        return _vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        return self.torch.from_numpy(a).to(self.device)

    def to_device_concat(self, arrays, pinned_limit=1 << 29):
        """The concatenation of 1-D float64 host arrays as one device tensor.  Up to ``pinned_limit`` bytes the pieces
        are copied straight into a pinned staging block (torch's caching host allocator: reused from call to call) and
        leave with one asynchronous DMA — one pass over the host data instead of np.concatenate's pass plus the
        runtime's chunked staging of a pageable source (64 x 10 s: ~10 ms instead of ~25)."""
        total = int(sum(len(a) for a in arrays))
        if total == 0 or total * 8 > pinned_limit:
            return self.to_device(np.concatenate(arrays) if len(arrays) else np.zeros(0))
        host = self.torch.empty((total,), dtype=self.torch.float64, pin_memory=True)
        view = host.numpy()
        offs = np.concatenate([[0], np.cumsum([len(a) for a in arrays])])

        def fill(lo, hi):
            for k in range(lo, hi):
                view[offs[k]:offs[k + 1]] = arrays[k]  # (NumPy releases the GIL for the copy)

        if total * 8 < (1 << 24) or len(arrays) < 8:
            fill(0, len(arrays))
            return host.to(self.device, non_blocking=True)
        # 82 MB of waveforms: ~9 ms on one host thread, ~2 on eight — and the DMA of a filled quarter (0.4 ms) runs under the
        # filling of the next one
        dev = self.torch.empty((total,), dtype=self.torch.float64, device=self.device)
        n = len(arrays)
        with _copy_pool() as pool:
            for c in range(4):
                a, b = n * c // 4, n * (c + 1) // 4
                cuts = [a + (b - a) * w // 8 for w in range(9)]
                list(pool.map(lambda w: fill(cuts[w], cuts[w + 1]), range(8)))
                dev[int(offs[a]):int(offs[b])].copy_(host[int(offs[a]):int(offs[b])], non_blocking=True)
        return dev

    def to_host(self, t, transpose=False):
        """Device tensor -> NumPy array.  Large results go through pinned host memory from torch's caching host
        allocator (a pageable destination makes the runtime stage the copy in small chunks: the 22 MB that one
        encode() of a 4.6 s utterance returns took ~7 ms that way, ~1 ms pinned); the array owns its pinned block and
        hands it back to the cache when it is dropped.  ``transpose``: return the reference's (bins, frames) layout of
        a frame-major [frames][bins] tensor, transposed on the device rather than by a strided host copy."""
        if transpose:
            t = t.transpose(0, 1).contiguous()
        if t.numel() * t.element_size() < (1 << 16):
            return t.cpu().numpy()
        host = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return host.numpy()

    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.torch.float64, device=self.device)

    @staticmethod
    def ptr(t):
        return _vp(t.data_ptr()) if t is not None else _vp(None)

    def profile(self, on=True):
        check(self.lib.wh_profile_enable(self.ctx, 1 if on else 0))

    def profile_collect(self, max_records=4096):
        """[(kernel name, ms), ...] in launch order since the last collect; synchronises the device."""
        names = ctypes.create_string_buffer(64 * max_records)
        ms = (ctypes.c_float * max_records)()
        n = _int(0)
        check(self.lib.wh_profile_collect(self.ctx, names, len(names), ms, max_records, ctypes.byref(n)))
        nm = names.value.decode().split("\n")[:n.value]
        return list(zip(nm, [float(ms[i]) for i in range(n.value)]))

    def take_flags(self):
        """Read-and-clear the sticky device condition flags (synchronises the current stream)."""
        buf = (ctypes.c_int32 * 16)()
        check(self.lib.wh_take_flags(self.ctx, self.stream(), buf))
        return list(buf)

    def post_flags(self, discard=False):
        """Enqueue the publication of the flags raised so far on this lane's stream (no host wait): wh_flags_post.
        ``discard``: clear them unpublished instead (conditions of work whose results nobody will use)."""
        check(self.lib.wh_flags_post(self.ctx, self.stream(), 1 if discard else 0))

    def poll_flags(self):
        """Conditions published by earlier ``post_flags`` calls that have executed, without synchronising."""
        buf = (ctypes.c_int32 * 16)()
        check(self.lib.wh_flags_poll(self.ctx, buf))
        return list(buf)

    def check_flags(self, where, allow=()):
        """take_flags() and raise WorldHipError for every condition that is set and not in ``allow``.  Returns the
        flag list (so that a caller can react to an allowed one, e.g. retry with a larger pulse capacity)."""
        return self.raise_for_flags(self.take_flags(), where, allow)

    @staticmethod
    def raise_for_flags(flags, where, allow=()):
        bad = [FLAG_MESSAGES.get(i, "device flag %d" % i) for i, v in enumerate(flags) if v and i not in allow]
        if bad:
            raise WorldHipError("%s: %s" % (where, "; ".join(bad)))
        return flags

    # ---- batch descriptor -----------------------------------------------------------------
    def make_batch(self, x_off, frame_off):
        return Batch(self, x_off, frame_off)


class Batch:
    def __init__(self, rt, x_off, frame_off):
        self.rt = rt
        self.x_off = np.ascontiguousarray(x_off, dtype=np.int64)
        self.frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
        self.n_utt = len(self.x_off) - 1
        h = _vp()
        check(rt.lib.wh_batch_create(rt.ctx, self.n_utt, self.x_off.ctypes.data_as(_c_i64p),
                                     self.frame_off.ctypes.data_as(_c_i64p), ctypes.byref(h)))
        self.handle = h

    @property
    def total_frames(self):
        return int(self.frame_off[-1])

    @property
    def total_samples(self):
        return int(self.x_off[-1])

    def __del__(self):
        try:
            if self.handle:
                self.rt.lib.wh_batch_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
