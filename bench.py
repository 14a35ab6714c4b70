#!/usr/bin/env python3
"""Benchmark of the MI355X WORLD hot path (BASELINE.json metric: analysis+synthesis frames/s and xRT,
16 kHz / 5 ms hop).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling, the default): BASELINE config 2 per GPU — 64 x 10 s synthetic 16 kHz utterances,
encode (DIO + StoneMask + CheapTrick + D4C) + decode (pulse-wise synthesis), inputs resident in HBM before the timed
region, outputs left in HBM.  A "step" is one full encode+decode pass over the rank's batch.  Utterances are
independent: ranks shard them with NO collective on the data path; the only communication is the barrier /
max-reduce of the timing.  `--scaling strong` fixes the TOTAL batch (`--utts` utterances) and shards it over the
ranks with world.distributed.shard_ranges (the product's sharding rule).

The timed region replays the step from a captured hipGraph when capture succeeds (`graph: true` in the JSON; the
same launches, no per-launch host work); the per-kernel HIP-event durations that feed `roofline` come from an
un-captured pass of the same K steps right after it (events cannot be read back from inside a graph).

`WH_BENCH_SHARE_GPU=1` (rehearsal only): every rank drives device 0 and the process group is gloo — the N > 1 code path
(per-rank input generation, sharding, barrier, max / sum reductions, `per_rank_ms`) runs end to end on a one-GPU lease;
the line says `"shared_gpu": true` and is NOT a scaling measurement.

Rank 0 prints ONE JSON line.  On a single GPU it also carries (each timed by this process):
  cpu_baseline   : the NumPy oracle on the host cores (1 core and all cores, median of 3), bounded sample;
  with_transfers : config 2 once more with the H2D of x and the D2H of f0/vuv/spectrogram/aperiodicity/out through
                   pinned buffers inside the timed region (what a host-buffer caller sees; never `value`);
  with_transfers_pipelined : the same with the results of step k leaving (DMA, private stream) under the upload (a
                   kernel reading the pinned waveform) and the kernels of step k+1; also reported at top level as
                   `value_with_transfers` — SURVEY §8(d) defines the metric with the API's transfers in it;
  config1_latency: the reference's own benchmark (test/speed.py:13-18): ONE World().encode(fs, x, 'harvest') and one
                   decode on test-mwm.wav through the drop-in facade, first call and warm, with the kernel share;
  other_configs  : BASELINE configs 3, 4 and 5 at their single-GPU sizes (ms per step, frames/s, heaviest kernels);
  feature_heads, swipe : the SURVEY §8(f) kernels on the config-2 batch (lfbank + mcep + imcep on the FP64 matrix
                   cores; f0_method='swipe');
  north_star     : BASELINE.json's target workload on ONE GPU — 1024 x 10 s, Harvest + CheapTrick + D4C-Requiem
                   encode + Requiem decode (>= 500 xRT asked).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "python-world_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FS = 16000
SECONDS = 10.0
UTT_PER_GPU = 64
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6  # FP64 vector FMA: 256 CUs x 4 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz (the SIMD-16 ceiling)


# Algorithmic (compulsory) HBM bytes per 5 ms frame of each dominant-kernel candidate, float64 API dtypes —
# SURVEY.md §8(d) components: x hop (640 B at 16 kHz), f0+vuv+tp 24 B, spectrogram and aperiodicity
# (fft/2+1)*8 B each (4104 B at fft 1024), output hop.
def algo_bytes_per_frame(fs, fft_size, out_hop_scale=1.0, requiem=False):
    """``requiem``: D4C-Requiem writes nap + 2 band values per frame instead of the (fft/2+1)-bin aperiodicity row and the
    Requiem decode reads them (24 B at 16 kHz, 56 B at 48 kHz)."""
    hop = int(fs * 5 // 1000) * 8
    kb = (fft_size // 2 + 1) * 8
    apb = (int(min(15000.0, fs / 2.0 - 3000.0) // 3000) + 2) * 8 if requiem else kb
    per_kernel = {
        "cheaptrick_kernel": hop + 24 + kb,
        "d4c_kernel": hop + 24 + apb,
        "req_filter_kernel": 24 + kb + apb + int(hop * out_hop_scale),
        "love_train_kernel": hop + 24 + 4,
        "response_kernel": 24 + kb + kb + int(hop * out_hop_scale),
        "stonemask_kernel": hop + 24 + 8,
        # Harvest kernels: the F0-only path reads the hop and writes f0/vuv/tp (664 B/frame)
        "hv_refine_kernel": hop + 24,
        "band_events_kernel": hop + 24,
    }
    # whole encode+decode path: 17744 B at 16 kHz; Requiem 9584 B
    path = hop + int(hop * out_hop_scale) + 48 + 2 * kb + 2 * apb
    return per_kernel, path


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=None,
                    help="utterances per GPU (weak) or in total (strong); default 64 = BASELINE config 2, 16 for config 5")
    ap.add_argument("--seconds", type=float, default=None, help="utterance length (default 10 s; 60 s for config 5)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=4, help="utterances per repeat of the 1-core CPU oracle leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the with_transfers and north_star blocks")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic (two short rocprofv3 --pmc child runs of this workload: FETCH_SIZE "
                         "and WRITE_SIZE in separate passes); the committed digest under profiles/ is quoted instead")
    ap.add_argument("--north-star-utts", type=int, default=1024)
    ap.add_argument("--transfer-lanes", type=int, default=0,
                    help="also time the with_transfers step dealt to this many lanes (copies of one lane under the "
                         "kernels of the next)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="independent sub-batches in flight per GPU, each on its own HIP stream (world.batch."
                         "WorldBatchLanes).  The default keeps one lane and clean per-kernel attribution")
    ap.add_argument("--no-stagger", action="store_true", help="start all lanes together (ablation)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="whole steps in flight per GPU: step k runs on pipeline k %% D (its own context, workspace, HIP "
                         "stream and hipGraph over its own resident copy of the batch), so that the latency-bound head of one "
                         "step (the DIO / Harvest serial kernels) runs under the chip-filling kernels of the step before.  "
                         "1 = one step at a time (`ms_per_step_one_in_flight` in the line either way)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 = DIO path encode+decode (the metric's config, default); "
                         "3 = Harvest F0 only; 4 = Harvest + CheapTrick + D4C-Requiem encode + Requiem decode; "
                         "5 = 48 kHz long-form: Harvest encode, scale_pitch(1.5), scale_duration(2.0), decode")
    args = ap.parse_args()
    if args.utts is None:
        args.utts = 16 if args.config == 5 else UTT_PER_GPU
    if args.seconds is None:
        args.seconds = 60.0 if args.config == 5 else SECONDS
    return args


# ---- synthetic input ------------------------------------------------------------------------------------------
def _synth_one(job):
    from world._synthetic import synth_utterance
    u, fs, seconds = job
    return synth_utterance(u, fs, seconds)


def make_inputs(first, count, fs, seconds, cache=True):
    """`count` distinct synthetic utterances (SURVEY §8(d) generator), generated on all host cores and cached in
    $WH_SYNTH_CACHE (default /tmp/wh_synth) so that repeated runs on one box do not pay for them again."""
    if count <= 0:
        return []
    cache_dir = os.environ.get("WH_SYNTH_CACHE", "/tmp/wh_synth")
    path = os.path.join(cache_dir, "u%d_n%d_fs%d_s%g.npy" % (first, count, fs, seconds))
    try:
        arr = np.load(path)
        return [arr[i] for i in range(count)]
    except Exception:
        pass
    jobs = [(first + i, fs, seconds) for i in range(count)]
    # under torch.distributed.run every rank generates its own utterances at the same time: share the host cores
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    procs = max(1, min(len(jobs), (os.cpu_count() or 1) // ranks_here, 32))
    if procs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            xs = pool.map(_synth_one, jobs)
    else:
        xs = [_synth_one(j) for j in jobs]
    if cache:
        try:
            os.makedirs(cache_dir, exist_ok=True)
            np.save(path, np.stack(xs))
        except Exception:
            pass
    return xs


# ---- CPU baseline (runs BEFORE the GPU is initialised: the all-cores leg forks) -------------------------------------
def _cpu_one(job):
    from oracle import api as oapi
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    x, fs, seed = job
    with ctx:
        dat = oapi.encode_np(fs, x, f0_method="dio")
        np.random.seed(seed)
        oapi.decode_np(dat)
    return len(dat["f0"])


def _cpu_one_north(job):
    from oracle import api as oapi
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    x, fs, _ = job
    with ctx:
        dat = oapi.encode_np(fs, x, f0_method="harvest", is_requiem=True)
        oapi.decode_np(dat)
    return len(dat["f0"])


def cpu_baseline_north(xs, fs, repeats=2):
    """The same-box CPU figure of the NORTH-STAR path: the NumPy oracle's encode(harvest,
    is_requiem=True) + Requiem decode, 1 core on one utterance and a process pool of one utterance per physical core;
    1 warm-up + `repeats` repeats each (about half a minute in all)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    _cpu_one_north((xs[0], fs, 0))
    one = []
    for r in range(repeats):
        t0 = time.perf_counter()
        frames = _cpu_one_north((xs[(r + 1) % len(xs)], fs, r))
        one.append(frames / (time.perf_counter() - t0))
    # a pool of 32 processes, one repeat: the Harvest oracle scales to ~5 x one core on this class of host whatever the pool
    # (3.0 k frames/s with 128 processes, memory-bound), and 128 utterances per repeat took 84 s
    pool_n = max(1, min(cores // 2, 32))
    jobs = [(xs[u % len(xs)], fs, u) for u in range(pool_n)]
    allc = []
    with mp.get_context("fork").Pool(pool_n) as pool:
        pool.map(_cpu_one_north, [(xs[0][:fs], fs, 0)] * pool_n)  # warm-up on a 1 s excerpt (imports; pocketfft keeps no plans)
        for r in range(1):
            t0 = time.perf_counter()
            frames = sum(pool.map(_cpu_one_north, jobs, chunksize=1))
            allc.append(frames / (time.perf_counter() - t0))
    v1, vn = float(np.median(one)), float(np.median(allc))
    return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "x_realtime": v1 * 0.005,
            "sample": "one 10 s utterance per repeat, encode(harvest, is_requiem=True)+decode through oracle/ (NumPy), "
                      "1 warm-up + median of %d repeats" % repeats,
            "all_cores": {"value": vn, "unit": "frames/s", "cores": pool_n, "host_cpus": cores, "x_realtime": vn * 0.005,
                          "sample": "%d utterances over a %d-process pool, one repeat (128 processes: 3.0 k frames/s, "
                                    "profiles/r06_bench_default_v1.json)" % (pool_n, pool_n)},
            "repeats_1core": [round(v, 1) for v in one], "repeats_all_cores": [round(v, 1) for v in allc]}


def cpu_baseline(xs, fs, n_utts, repeats=3):
    """The NumPy oracle (a 'port' of the reference path) on a bounded sample of the same workload:
    (i) 1 core, per-utterance loop — the reference's execution model; (ii) all host cores, a process pool over
    utterances.  One warm-up utterance, then `repeats` timed repeats each; medians reported."""
    import multiprocessing as mp

    n_utts = max(1, min(n_utts, len(xs)))
    cores = os.cpu_count() or 1
    _cpu_one((xs[0], fs, 0))  # warm-up (imports, FFT plans)
    one = []
    for r in range(repeats):
        t0 = time.perf_counter()
        frames = sum(_cpu_one((xs[u], fs, u)) for u in range(n_utts))
        one.append(frames / (time.perf_counter() - t0))
    # one worker per PHYSICAL core (os.cpu_count() counts SMT threads; 256 workers on this 2 x 64-core box ran at 7.8 k
    # frames/s against 13 k with 64: the oracle's FFTs and gathers are memory-bound)
    pool_n = max(1, cores // 2)
    jobs = [(xs[u % len(xs)], fs, u) for u in range(pool_n)]
    allc = []
    t_pool = time.perf_counter()
    with mp.get_context("fork").Pool(pool_n) as pool:
        # warm-up inside the workers on 1 s excerpts (imports; pocketfft keeps no plans), then two repeats: this leg was
        # 100 of the 126 s the CPU baseline took of a default run
        pool.map(_cpu_one, [(xs[0][:fs], fs, 0)] * pool_n)
        for r in range(min(repeats, 2)):
            t0 = time.perf_counter()
            frames = sum(pool.map(_cpu_one, jobs, chunksize=1))
            allc.append(frames / (time.perf_counter() - t0))
    t_pool = time.perf_counter() - t_pool
    v1, vn = float(np.median(one)), float(np.median(allc))
    sec = len(xs[0]) / fs
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d utterances (%.0f s each) per repeat, encode(dio)+decode through oracle/ (NumPy), "
                      "1 warm-up + median of %d repeats" % (n_utts, len(xs), sec, repeats),
            "x_realtime": v1 * 0.005,
            "all_cores": {"value": vn, "unit": "frames/s", "cores": pool_n, "host_cpus": cores, "x_realtime": vn * 0.005,
                          "sample": "%d utterances per repeat over a %d-process pool, median of %d repeats"
                                    % (pool_n, pool_n, min(repeats, 2))},
            # the reference ITSELF (not this port), measured once in the authoring container (8 vCPU
            # Xeon 2.1 GHz, numba absent): sum of its five stages on one 10 s / 16 kHz utterance = 5.14 s
            "reference_measured": {"value": 389.0, "unit": "frames/s", "cores": 1, "x_realtime": 1.95,
                                   "source": "BASELINE.md section 2 (authoring container, not this box)"},
            "cpu_model": model, "repeats_1core": [round(v, 1) for v in one], "repeats_all_cores": [round(v, 1) for v in allc]}


def pmc_table(name):
    """rows of a committed rocprofv3 PMC digest under profiles/ (bench.py cannot run the profiler around itself)."""
    path = os.path.join(ROOT, "profiles", name)
    rows = {}
    try:
        for line in open(path):
            parts = line.split()
            if parts and not line.startswith("#") and parts[0].endswith("_kernel"):
                rows[parts[0]] = parts
    except Exception:
        pass
    return rows, os.path.relpath(path, ROOT)


def pmc_traffic(kernel, lanes, config):
    rows, src = pmc_table("hbm_traffic_cfg%d_latest.txt" % config)
    if kernel in rows:
        return float(rows[kernel][3]) * 1e6 / lanes, src
    return None, None


SQ_PASS = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64",
           "SQ_INSTS_VALU_TRANS_F64", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVES")
LAST_SQ = {}  # per kernel: the SQ counters of the latest measure_pmc_traffic call (third child run), averaged per launch


def fp64_view(kernel, avg_launch_ms):
    """The FP64-issue view of one kernel from the SQ pass of the latest measure_pmc_traffic call (the
    honest roof of this path is FP64 vector issue, not HBM): FLOPs per launch = (2 FMA + ADD + MUL + TRANS) x 64 lanes over
    the launch duration THIS run measured with HIP events; the VALU's share of a resident wave's lifetime; the share of
    the VALU instructions that are FP64 arithmetic.  None when the pass did not run."""
    c = LAST_SQ.get(kernel)
    if not c or not avg_launch_ms:
        return None
    f64 = sum(c.get(k, 0.0) for k in SQ_PASS[1:5])
    flops = (f64 + c.get("SQ_INSTS_VALU_FMA_F64", 0.0)) * 64.0
    tf = flops / (avg_launch_ms * 1e-3) / 1e12
    return {"fp64_TFLOPs": tf, "fp64_peak_TFLOPs": FP64_VECTOR_PEAK_TFLOPS, "fp64_peak_frac": tf / FP64_VECTOR_PEAK_TFLOPS,
            "valu_busy": c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0)),
            "fp64_inst_share": f64 / max(1.0, c.get("SQ_INSTS_VALU", 0.0)), "flops_per_launch": flops,
            "note": "valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: the VALU's share of ONE resident wave's lifetime (a "
                    "SIMD holds 3-6 such waves); counters from a third rocprofv3 --pmc child run of this workload, "
                    "duration from this run's HIP events"}


def measure_pmc_traffic(args, timeout_s=240, config=None, utts=None, seconds=None, steps=2, sq=True):
    """HBM bytes per launch of every kernel of THIS workload, measured now: two child runs of this script under
    `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE and WRITE_SIZE cannot share a pass; nothing but the kernel
    trace beside the counters), corrected as MI355X_MICROARCH.md prescribes for gfx950: (2*FETCH_SIZE + WRITE_SIZE) KiB.
    Returns ({kernel: bytes per launch}, description) or (None, reason)."""
    import collections
    import csv
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="wh_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    child = [sys.executable, os.path.abspath(__file__), "--config", str(config or args.config),
             "--utts", str(utts or args.utts), "--seconds", str(seconds or args.seconds), "--steps", str(steps),
             "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-graph", "--no-pmc", "--in-flight", "1"]
    per = {}
    try:
        LAST_SQ.clear()
        for counter in ("FETCH_SIZE", "WRITE_SIZE") + (("SQ",) if sq else ()):
            d = os.path.join(tmp, counter)
            names = list(SQ_PASS) if counter == "SQ" else [counter]
            cmd = [exe, "--pmc"] + names + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--"] + child
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=tmp)
            path = None
            for dirpath, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(dirpath, f)
            if counter == "SQ":  # the FP64-issue view: optional, never costs the traffic figures
                if r.returncode == 0 and path is not None:
                    tot2 = collections.defaultdict(lambda: collections.defaultdict(float))
                    n2 = collections.defaultdict(collections.Counter)
                    for row in csv.DictReader(open(path)):
                        mt = re.search(r"(\w+_kernel)\b", row["Kernel_Name"])
                        if not mt or "at::native" in row["Kernel_Name"]:
                            continue
                        tot2[mt.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
                        n2[mt.group(1)][row["Counter_Name"]] += 1
                    for k, cs in tot2.items():
                        LAST_SQ[k] = {c: v / n2[k][c] for c, v in cs.items()}
                continue
            if r.returncode != 0 or path is None:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-300:])
            tot, n = collections.defaultdict(float), collections.Counter()
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] != counter:
                    continue
                mt = re.search(r"(\w+_kernel)\b", row["Kernel_Name"])
                if not mt or "at::native" in row["Kernel_Name"]:
                    continue
                tot[mt.group(1)] += float(row["Counter_Value"])
                n[mt.group(1)] += 1
            per[counter] = {k: tot[k] / n[k] for k in tot}
    except Exception as e:  # a profiler problem must never cost the headline line
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {k: (2.0 * f + per["WRITE_SIZE"].get(k, 0.0)) * 1024.0 for k, f in per["FETCH_SIZE"].items()}
    # a few HIP-event timer names (what kernel_ms is keyed by) cover a kernel whose symbol carries a variant suffix:
    # "band_events_kernel" times band_events_ols_kernel<1|4>
    for table in (out, LAST_SQ):
        for sym in list(table):
            stem = sym[:-len("_kernel")]
            for cut in ("_ols", "_tab"):
                if stem.endswith(cut):
                    table.setdefault(stem[:-len(cut)] + "_kernel", table[sym])
    return out, ("measured in this run: two child runs of this workload under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in "
                 "separate passes, --kernel-trace only), bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB averaged per launch "
                 "(gfx950 correction of MI355X_MICROARCH.md)")


def pmc_fp64_flops(kernel, lanes, config):
    rows, src = pmc_table("sq_counters_cfg%d_latest.txt" % config)
    if kernel in rows:
        return float(rows[kernel][1]) * 1e-6 * float(rows[kernel][-1]) * 1e12 / lanes, src
    return None, None


SECTION_S = {}
T_START = time.perf_counter()


def main():
    global FS
    args = parse()
    if args.config == 5:
        FS = 48000
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))

    # ---- inputs (host) and the CPU baseline, before any GPU state exists in this process --------------------------
    if args.scaling == "strong":
        from world.distributed import shard_ranges
        lo, hi = shard_ranges([int(FS * args.seconds)] * args.utts, world)[rank]
        first, count = lo, hi - lo
    else:
        first, count = rank * args.utts, args.utts
    distinct = min(count, 64 if args.config != 5 else 16)  # a larger batch repeats its first 64 (16) utterances
    xs_distinct = make_inputs(first, distinct, FS, args.seconds)
    xs = [xs_distinct[i % distinct] for i in range(count)]
    cpu = None
    SECTION_S["inputs"] = round(time.perf_counter() - T_START, 1)
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.config == 2:
        t_blk = time.perf_counter()
        cpu = cpu_baseline(xs_distinct, FS, args.cpu_utts)
        SECTION_S["cpu_baseline"] = round(time.perf_counter() - t_blk, 1)
    args.north_star_cpu = None
    if cpu is not None and not args.no_extras and args.scaling == "weak":
        t_blk = time.perf_counter()
        try:
            args.north_star_cpu = cpu_baseline_north(xs_distinct, FS)
        except Exception as e:  # never costs the headline
            args.north_star_cpu = {"error": "%s: %s" % (type(e).__name__, e)}
        SECTION_S["cpu_baseline_north_star"] = round(time.perf_counter() - t_blk, 1)
    xs_cfg5 = None
    xs_north = None
    if world == 1 and not args.no_extras and args.config == 2 and args.scaling == "weak":
        try:  # the long-form inputs of the other_configs block are generated here, before any GPU state exists (fork)
            xs_cfg5 = make_inputs(0, 16, 48000, 60.0)
        except Exception:
            xs_cfg5 = None
        try:  # the north-star batch on DISTINCT utterances (1.3 GB of host memory; ~10 s on 32 cores, not cached on disk)
            if (os.cpu_count() or 1) >= 8:
                xs_north = xs_distinct + make_inputs(first + distinct, args.north_star_utts - distinct, FS, args.seconds,
                                                     cache=False)
        except Exception:
            xs_north = None
        SECTION_S["inputs_all"] = round(time.perf_counter() - T_START, 1)

    import torch
    import torch.distributed as dist

    # WH_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on ONE GPU — every rank on device 0, gloo for the barrier and the
    # reductions (two RCCL ranks cannot share a device); never a scaling number
    share_gpu = os.environ.get("WH_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run the RCCL path is exercised even for 1 rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if share_gpu else "cuda"  # where the timing / count reductions live (gloo reduces host tensors)

    from world.batch import WorldBatchLanes

    # args.lanes independent sub-batches per GPU, each on its own HIP stream and library context; one "step" is
    # still one pass of the hot path over the whole per-GPU batch
    n_lanes = max(1, min(args.lanes, max(count, 1)))
    depth = max(1, args.in_flight) if n_lanes == 1 else 1
    # `depth` independent pipelines over the SAME batch (each with its own resident copy of the inputs, context,
    # workspace arena and stream): step k of the timed region runs on pipeline k % depth.  Nothing is shared and
    # nothing is skipped — every step is a full pass over the whole per-GPU batch; what two steps in flight buy is that
    # the serial, latency-bound kernels at the head of a step (decimation IIRs, contour tracking: a few dozen
    # workgroups) run under the chip-filling kernels at the tail of the step before instead of on an idle chip.
    wls = [WorldBatchLanes(local_rank, lanes=n_lanes, first_lane=(d + 1) if depth > 1 else None) for d in range(depth)]
    wl = wls[0]
    rts = [wb.rt for wb in wl.lanes]
    for w_ in wls:
        w_.upload(xs, FS)  # inputs resident in HBM before the timed region
    frames_per_step = wl.total_frames

    steps_fn = [make_step(args, w_, FS) for w_ in wls]
    step = steps_fn[0]

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # (--warmup 0 still primes every pipeline once: the one-off table uploads and arena allocations of a context's first
    # call can be neither captured into a hipGraph nor timed as a step; the line says so: "primed")
    for w in range(max(args.warmup, 1)):
        for fn in steps_fn:
            fn(w)
    fence()
    for w_ in wls:
        for wb in w_.lanes:
            wb.check("bench warm-up")

    # ---- hipGraph capture of one step per pipeline (single lane each; falls back to eager launches if anything refuses) --
    graphs = None
    if not args.no_graph and n_lanes == 1:
        graphs = [try_capture(torch, (lambda fn=fn: fn(1000)), stream=w_.lanes[0].rt.own_stream) for fn, w_ in zip(steps_fn, wls)]
        if any(g is None for g in graphs):
            graphs = None
    graph = graphs[0] if graphs else None
    pipe_streams = [w_.lanes[0].rt.own_stream for w_ in wls]
    fence()

    def run_steps(active):
        """K steps dealt round-robin to the first `active` pipelines; returns (host enqueue time, wall time)."""
        t0 = time.perf_counter()
        for k in range(args.steps):
            d = k % active
            if graphs is not None:
                if pipe_streams[d] is not None:
                    with torch.cuda.stream(pipe_streams[d]):
                        graphs[d].replay()
                else:
                    graphs[d].replay()
            else:
                steps_fn[d](1000 + k)
        enq = time.perf_counter() - t0  # the launches are asynchronous: host time to enqueue all K steps
        fence()
        return enq, time.perf_counter() - t0

    host_enqueue, elapsed = run_steps(depth)
    one_in_flight = None
    if depth > 1:  # the same K steps with one step in flight at a time (pipeline 0 alone), for the record
        fence()
        _, one_in_flight = run_steps(1)

    # ---- un-captured pass of the same K steps with a HIP-event pair around every kernel launch ------------------
    for rt in rts:
        rt.profile(True)
    tp0 = time.perf_counter()
    for k in range(args.steps):
        step(1000 + k)
    fence()
    eager_elapsed = time.perf_counter() - tp0
    records, flags = [], [0] * 16
    for rt in rts:
        records += rt.profile_collect()
        rt.profile(False)
        flags = [a | b for a, b in zip(flags, rt.take_flags())]

    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist.is_initialized():
        mine = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # every rank's own clock: a straggler shows in the line, not only in the max
        per_rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([frames_per_step, count], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        frames_all, utts_all = float(tot[0].item()), float(tot[1].item())
    else:
        frames_all, utts_all = float(frames_per_step), float(count)

    if rank == 0:
        total_frames = frames_all * args.steps
        audio_s = utts_all * args.seconds * args.steps
        value = total_frames / elapsed
        # per-kernel launch durations from the HIP-event records of the profiled pass (rank 0, each lane's stream)
        agg = {}
        for name, ms in records:
            a = agg.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        kernel_ms = {k: v[0] / v[1] for k, v in agg.items()}
        dominant = max(agg.items(), key=lambda kv: kv[1][0])[0] if agg else None
        roofline = None
        if dominant:
            from world.cheaptrick import default_fft_size
            per_k, path_b = algo_bytes_per_frame(FS, default_fft_size(FS), 2.0 if args.config == 5 else 1.0,
                                                 requiem=args.config == 4)
            per_frame = per_k.get(dominant, path_b)
            avg_s = kernel_ms[dominant] / 1e3
            # every lane launches the kernel once per step on its share of the frames
            frames_per_launch = frames_per_step * args.steps / agg[dominant][1]
            achieved = per_frame * frames_per_launch / avg_s / 1e9
            std = args.utts == UTT_PER_GPU and args.scaling == "weak" and args.config in (2, 3, 4)
            traffic, traffic_src, traffic_all = None, None, None
            if world == 1 and not args.no_pmc and len(rts) == 1:
                measured, how = measure_pmc_traffic(args)
                if measured and dominant in measured:
                    traffic, traffic_src, traffic_all = measured[dominant], {"source": how}, measured
                else:
                    traffic_src = {"source": "measurement unavailable", "reason": how if not measured else "kernel not in the trace"}
            if traffic is None and std:
                traffic, f = pmc_traffic(dominant, len(rts), args.config)
                if traffic is not None:
                    traffic_src = {"source": "committed profile", "file": f, "fallback_because": traffic_src,
                                   "note": "rocprofv3 --pmc passes of this workload run by the builder; not a measurement "
                                           "of this run"}
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": traffic_src,
                        "traffic_over_algorithmic": None if traffic is None else traffic / (per_frame * frames_per_launch),
                        "traffic_per_kernel_MB": None if traffic_all is None else
                        {k: round(v / 1e6, 1) for k, v in sorted(traffic_all.items(), key=lambda kv: -kv[1])[:8]},
                        "algorithmic_bytes_per_launch": per_frame * frames_per_launch,
                        "frames_per_launch": frames_per_launch,
                        "avg_launch_ms": kernel_ms[dominant],
                        "timing": "HIP events around every launch, un-captured pass of the same %d steps" % args.steps,
                        "path_algorithmic_GBps": path_b * frames_per_step * args.steps / elapsed / 1e9}
            # second view of the same kernel: the path is FP64-compute/latency-bound, not HBM-bound —
            # measured in this run when the PMC child runs ran (fp64_TFLOPs, fp64_peak_frac, valu_busy), else from the
            # committed digest
            live = fp64_view(dominant, kernel_ms[dominant]) if traffic_all is not None else None
            if live:
                roofline.update({k: live[k] for k in ("fp64_TFLOPs", "fp64_peak_frac", "valu_busy", "fp64_inst_share")})
                roofline["fp64_vector"] = dict(live, achieved=live["fp64_TFLOPs"], peak=FP64_VECTOR_PEAK_TFLOPS, unit="TFLOP/s",
                                               frac=live["fp64_peak_frac"], flops_source={"source": "measured in this run"})
                roofline["fp64_per_kernel"] = {k: {kk: round(v[kk], 4) for kk in ("fp64_TFLOPs", "fp64_peak_frac", "valu_busy", "fp64_inst_share")}
                                               for k, v in ((k, fp64_view(k, kernel_ms.get(k)))
                                                            for k in sorted(kernel_ms, key=lambda q: -kernel_ms[q])[:6]) if v}
            flops, flops_src = pmc_fp64_flops(dominant, len(rts), args.config) if (std and not live) else (None, None)
            if flops:
                roofline["fp64_vector"] = {"achieved": flops / avg_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                           "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                           "flops_per_launch": flops,
                                           "flops_source": {"source": "committed profile", "file": flops_src}}
        noise_note = "on-device Philox noise (not the reference-parity host-noise path)"
        workloads = {
            2: "BASELINE config 2 per GPU: %d x %.0f s synthetic 16 kHz utterances, DIO+StoneMask+CheapTrick+D4C encode + "
               "pulse-wise synthesis decode with " + noise_note + ", HBM-resident",
            3: "BASELINE config 3 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest F0 only, HBM-resident",
            4: "BASELINE config 4 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest+CheapTrick+D4C-Requiem encode "
               "+ Requiem decode, HBM-resident",
            5: "BASELINE config 5 per GPU: %d x %.0f s synthetic 48 kHz utterances, Harvest+CheapTrick+D4C encode, "
               "scale_pitch(1.5), scale_duration(2.0), pulse-wise decode with " + noise_note + ", HBM-resident"}
        out = {
            "metric": "analysis+synthesis frames/sec (and xRT), %d kHz / 5 ms hop" % (FS // 1000),
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "steps_in_flight": depth,
            "headline_definition": "ms_per_step / value: %d whole step(s) in flight (the default since round 5; rounds 1-4 "
                                   "timed one: that figure is ms_per_step_one_in_flight in every line)" % depth,
            "primed": args.warmup == 0,
            "ms_per_step_one_in_flight": None if one_in_flight is None else one_in_flight / args.steps * 1e3,
            "per_rank_ms": [round(v, 4) for v in per_rank_ms],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "x_realtime": audio_s / elapsed,
            "graph": graph is not None,
            "eager_ms_per_step": eager_elapsed / args.steps * 1e3,
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
            "config": {"workload": workloads[args.config] % (count, args.seconds),
                       "utterances_per_gpu": count, "distinct_utterances": distinct, "fs": FS, "frame_period_ms": 5,
                       "frames_per_step_per_gpu": frames_per_step, "sharding": "utterances, no collective",
                       "lanes_per_gpu": len(rts)},
            "roofline": roofline,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])},
            "device_flags": flags,
        }
        if share_gpu:
            out["shared_gpu"] = True
            out["config"]["parallelism"] = "%d ranks time-sharing ONE GPU over gloo (WH_BENCH_SHARE_GPU=1): a rehearsal " \
                                           "of the multi-rank path, not a scaling measurement" % world
        if cpu is not None:
            out["cpu_baseline"] = cpu
        SECTION_S["headline_done_at"] = round(time.perf_counter() - T_START, 1)
        if world == 1 and not args.no_extras and args.config == 2 and args.scaling == "weak":
            graph = graphs = None  # (the graphs' private pools go back to the allocator)
            if depth > 1:
                # the blocks below drive ONE pipeline from torch's current stream (copies and kernels in one order): lane 0
                del steps_fn, step
                for w_ in wls:
                    w_.resident = None
                wl = WorldBatchLanes(local_rank, lanes=1)
                wl.upload(xs, FS)
                torch.cuda.empty_cache()
            blocks = [("with_transfers", lambda: with_transfers_block(torch, wl, xs, FS)),
                      ("with_transfers_pipelined", lambda: with_transfers_pipelined_block(torch, wl, xs, FS)),
                      ("roundtrip_out_only", lambda: roundtrip_out_only_block(torch, wl, xs, FS))]
            if args.transfer_lanes > 1:  # measured 41 ms with 4 lanes against 36.5 ms serial: off by default
                blocks.append(("with_transfers_overlapped",
                               lambda: with_transfers_lanes_block(torch, local_rank, xs, FS, args.transfer_lanes)))
            blocks.append(("varying_lengths", lambda: varying_lengths_block(torch, local_rank, xs, FS)))
            blocks.append(("decode_alone", lambda: decode_alone_block(torch, wl, FS)))
            blocks.append(("facade_batch", lambda: facade_batch_block(torch, xs, FS)))
            blocks.append(("config1_latency", lambda: config1_latency_block(torch)))
            blocks.append(("feature_heads", lambda: feature_heads_block(torch, wl, FS)))
            blocks.append(("swipe", lambda: swipe_block(torch, wl, FS)))
            blocks.append(("other_configs", lambda: other_configs_block(torch, local_rank, xs_distinct, xs_cfg5, max(1, args.in_flight))))
            for key, fn in blocks + [("north_star", lambda: north_star_block(torch, local_rank, xs_north or xs_distinct, FS, args))]:
                t_blk = time.perf_counter()
                try:
                    out[key] = fn()
                except Exception as e:  # an extra block must never cost the headline line
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                SECTION_S[key] = round(time.perf_counter() - t_blk, 1)
            SECTION_S["total"] = round(time.perf_counter() - T_START, 1)
            out["section_seconds"] = dict(SECTION_S)  # where the wall time of this run went (host clock)
            piped = out.get("with_transfers_pipelined", {})
            if "value" in piped:  # SURVEY §8(d) states the metric with the API's H2D / D2H inside
                out["value_with_transfers"] = piped["value"]
                out["ms_per_step_with_transfers"] = piped["ms_per_step"]
                # the figure SURVEY §8(d) words the metric on (wall time incl. H2D of x and D2H of every API output)
                out["value_survey_8d"] = piped["value"]
            if "value" in out.get("roundtrip_out_only", {}):
                out["value_roundtrip_out_only"] = out["roundtrip_out_only"]["value"]
            # `value` is the HBM-resident rate (the bench contract: inputs resident when the timed region starts); SURVEY
            # §8(d) words the metric with the API's transfers inside: that figure is value_with_transfers (all tensors)
            # and value_roundtrip_out_only (audio only) — never quote `value` as the host-buffer rate
            out["value_resident"] = out["value"]
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def make_step(args, wl, fs):
    if args.config == 2:
        def step(seed):
            encs = wl.encode_device(fs, stagger=not args.no_stagger, f0_method="dio")
            return wl.decode_device(encs, seed=seed)
    elif args.config == 3:
        from world.harvest import harvest_device

        def step(seed):
            out = []
            for wb, r in zip(wl.lanes, wl.resident):
                with wb.rt.on_stream():
                    out.append(harvest_device(wb.rt, r[0], r[1], r[2], fs))
            return out
    elif args.config == 5:
        def step(seed):
            encs = wl.encode_device(fs, stagger=not args.no_stagger, f0_method="harvest")
            for e in encs:
                with e.rt.on_stream():
                    e.scale_pitch(1.5).scale_duration(2.0)
            return wl.decode_device(encs, seed=seed)
    else:
        def step(seed):
            encs = wl.encode_device(fs, stagger=not args.no_stagger, f0_method="harvest", is_requiem=True)
            return wl.decode_device(encs)  # seed tables: generated on the device once (wh_requiem_seeds), resident
    return step


def try_capture(torch, fn, stream=None):
    """Capture one call of `fn` (kernel launches on torch's current stream, allocations from torch's graph pool) into
    a hipGraph.  None if capture is refused (a synchronous call inside the step, an unsupported node...).
    ``stream``: capture on this stream (a pipeline whose calls launch on a private stream of its own)."""
    try:
        g = torch.cuda.CUDAGraph()
        s = stream if stream is not None else torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                keep = fn()
        g._keep = keep  # outputs live in the graph's private pool
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.stream(s):
            g.replay()
        torch.cuda.synchronize()
        return g
    except Exception as e:
        sys.stderr.write("hipGraph capture failed, timing eager launches: %s: %s\n" % (type(e).__name__, e))
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return None


def with_transfers_block(torch, wl, xs, fs, steps=5):
    """Config 2 as a host-buffer caller sees it: per step H2D of the waveforms from pinned memory, encode + decode,
    D2H of f0 / vuv / spectrogram / aperiodicity / out into pinned buffers, one stream, nothing overlapped."""
    wb = wl.lanes[0]
    rt = wb.rt
    batch, x_d, tp_d = wl.resident[0]
    x_pin = torch.from_numpy(np.concatenate(xs)).pin_memory()
    enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
    y, _ = wb.decode_device(enc, seed=1, check=False)
    outs = [enc.f0, enc.vuv, enc.spectrogram, enc.aperiodicity, y]
    pins = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in outs]
    nbytes_d2h = sum(t.numel() * t.element_size() for t in outs)

    def one(seed):
        x_d.copy_(x_pin, non_blocking=True)
        e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
        yy, _ = wb.decode_device(e, seed=seed, check=False)
        for p, t in zip(pins, (e.f0, e.vuv, e.spectrogram, e.aperiodicity, yy)):
            p.copy_(t, non_blocking=True)

    one(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(10 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    wb.check("with_transfers")
    frames = batch.total_frames
    return {"ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s",
            "x_realtime": len(xs) * len(xs[0]) / fs / dt, "steps": steps,
            "h2d_MB_per_step": x_pin.numel() * 8 / 1e6, "d2h_MB_per_step": nbytes_d2h / 1e6,
            "note": "config 2 step incl. H2D of x and D2H of f0/vuv/spectrogram/aperiodicity/out via pinned host "
                    "buffers on one stream (no overlap); 'ps spectrogram' is not materialised"}


def with_transfers_pipelined_block(torch, wl, xs, fs, steps=6):
    """The host-buffer step as a streaming caller runs it (WorldBatch.download_async): the D2H of step k's results
    goes out on a second stream (its own copy engine) while step k+1 is uploaded and computed — results
    double-buffered in pinned memory.  The steady state is the slower of (H2D + kernels) and D2H, not their sum."""
    wb = wl.lanes[0]
    batch, x_d, tp_d = wl.resident[0]
    x_pin = torch.from_numpy(np.concatenate(xs)).pin_memory()

    def one(k):
        # the upload is a kernel reading the pinned buffer, not a DMA copy: on the DMA queue it can land behind the
        # previous step's 1.1 GB download and stall the whole step (tools/pipe_variants.py: 23.6 ms in 9/9 runs
        # against 23.9-26.1 — and 33-36 ms on some boxes — with both directions on the copy engines)
        wb.refill_from_pinned(x_d, x_pin)
        e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
        yy, _ = wb.decode_device(e, seed=10 + k, check=False)
        return wb.download_async((e.f0, e.vuv, e.spectrogram, e.aperiodicity, yy), slot=k % 2)

    one(0)
    one(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(2 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rounds = []
    for r in range(4):  # the same loop four more times: the figure must not be a lucky mode
        t1 = time.perf_counter()
        for k in range(steps):
            one(2 + k)
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t1) / steps * 1e3)
    wb.check("with_transfers_pipelined")
    frames = batch.total_frames
    return {"ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s",
            "x_realtime": len(xs) * len(xs[0]) / fs / dt, "steps": steps,
            "repeat_rounds_ms_per_step": [round(v, 2) for v in rounds],
            "upload": "kernel reading the pinned host buffer (wh_copy_mapped)", "download": "DMA, private stream",
            "note": "with_transfers with the D2H of one step's results on a second stream under the upload + kernels "
                    "of the next (WorldBatch.download_async, double-buffered pinned results): throughput of a "
                    "streaming host-buffer caller"}


def roundtrip_out_only_block(torch, wl, xs, fs, steps=6):
    """The round trip of a caller that wants the AUDIO back and nothing else (encode -> decode on the device, the
    analysis tensors never leave HBM): per step H2D of the waveforms (a kernel reading the pinned buffer) and D2H of
    `out` alone (82 + 82 MB at config 2, against 1 135 MB when the dense spectra are downloaded too), the download of
    step k under the upload + kernels of step k+1."""
    wb = wl.lanes[0]
    batch, x_d, tp_d = wl.resident[0]
    x_pin = torch.from_numpy(np.concatenate(xs)).pin_memory()

    def one(k):
        wb.refill_from_pinned(x_d, x_pin)
        e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check='deferred')
        yy, _ = wb.decode_device(e, seed=10 + k, check='deferred')
        return wb.download_async((yy,), slot=k % 2), yy

    one(0)
    (_, _), y0 = one(1)
    torch.cuda.synchronize()
    rounds = []
    for r in range(4):
        t1 = time.perf_counter()
        for k in range(steps):
            one(2 + k)
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t1) / steps)
    dt = float(np.median(rounds))
    wb.check("roundtrip_out_only")
    frames = batch.total_frames
    return {"ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s",
            "x_realtime": len(xs) * len(xs[0]) / fs / dt, "steps": steps,
            "rounds_ms_per_step": [round(v * 1e3, 2) for v in rounds],
            "h2d_MB_per_step": x_pin.numel() * 8 / 1e6, "d2h_MB_per_step": y0.numel() * 8 / 1e6,
            "flag_check": "deferred (wh_flags_post / wh_flags_poll: no host wait per call)",
            "note": "config 2 with H2D of x and D2H of `out` only inside the timed region, pipelined: what a "
                    "resynthesis caller (encode -> modify -> decode) pays; median of 4 rounds"}


def decode_alone_block(torch, wl, fs, reps=10):
    """The decode of config 2 on its own: (a) with the time base that encode prefetched under CheapTrick / D4C (only the
    spectral half runs), (b) with that time base dropped, i.e. what decode_batch of a fresh from_dicts encoding or a
    decode after scale_pitch / scale_duration pays: prep, the exact phase scan, the pulse kernels, then the responses.
    HIP-event times of whole calls; per-kernel times of (b) from one profiled call."""
    wb = wl.lanes[0]
    rt = wb.rt
    batch, x_d, tp_d = wl.resident[0]
    enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
    wb.decode_device(enc, seed=1, check=False)
    ms_pref, _ = _event_ms(torch, lambda: wb.decode_device(enc, seed=1, check=False), reps)
    tb, enc._timebase = enc._timebase, None
    wb.decode_device(enc, seed=1, check=False)
    ms_inline, _ = _event_ms(torch, lambda: wb.decode_device(enc, seed=1, check=False), reps)
    rt.profile(True)
    wb.decode_device(enc, seed=1, check=False)
    agg = {}
    for name, t in rt.profile_collect():
        agg[name] = agg.get(name, 0.0) + t
    rt.profile(False)
    enc._timebase = tb
    wb.check("decode_alone")
    return {"prefetched_timebase_ms": ms_pref, "inline_timebase_ms": ms_inline, "timebase_cost_ms": ms_inline - ms_pref,
            "kernel_ms_inline": {k: round(v, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]},
            "note": "config-2 encoding, decode only; 'inline' = no prefetched time base (fresh from_dicts encodings, "
                    "modified encodings)"}


def facade_batch_block(torch, xs, fs, reps=3):
    """The public batched API with HOST arrays on both sides, two callers:
    (a) `resynthesis`: World().encode_batch -> scale_pitch -> scale_duration -> decode_batch, the reference's prosody
        flow (example/prosody.py:38-57).  encode_batch returns lazy dicts (world.batch.EncodingDict): the dense tensors
        that the caller never reads stay in HBM, so the waveforms, the per-frame scalars and the audio cross PCIe and
        nothing else;
    (b) `materialised`: the same two calls by a caller that reads every dense value (dict(d) per utterance: every
        tensor downloaded and transposed to the reference's (bins, frames) layout, and uploaded again by decode_batch)
        — what the eager dicts of the earlier rounds cost every caller."""
    from world import main

    W = main.World()

    def resynthesis(copy_out=True):
        dats = W.encode_batch(fs, xs, f0_method="dio")
        for d in dats:
            W.scale_pitch(d, 1.5)
            W.scale_duration(d, 2.0)
        return W.decode_batch(dats, copy_out=copy_out)

    def materialised():
        t0 = time.perf_counter()
        dats = [dict(d) for d in W.encode_batch(fs, xs, f0_method="dio")]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        W.decode_batch(dats)
        torch.cuda.synchronize()
        return dats, t1 - t0, time.perf_counter() - t1

    def roundtrip(copy_out=True):  # no modification: directly comparable with the resident step and with round 4's 46 + 40 ms
        return W.decode_batch(W.encode_batch(fs, xs, f0_method="dio"), copy_out=copy_out)

    resynthesis()
    materialised()
    roundtrip()
    torch.cuda.synchronize()
    flow_s, enc_s, dec_s, rt_s = [], [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        roundtrip()
        torch.cuda.synchronize()
        rt_s.append(time.perf_counter() - t0)
    for _ in range(reps):
        t0 = time.perf_counter()
        dats = resynthesis()
        torch.cuda.synchronize()
        flow_s.append(time.perf_counter() - t0)
    frames = sum(len(d["f0"]) for d in dats)
    for _ in range(reps):
        _, e, d = materialised()
        enc_s.append(e)
        dec_s.append(d)
    # the same flow as ONE batch (main.FACADE_SPLIT_BYTES: by default a batch of this size runs as two parts on two
    # pipelines, one part's PCIe transfer under the other's kernels)
    split_bytes, single_s = main.FACADE_SPLIT_BYTES, []
    main.FACADE_SPLIT_BYTES = 1 << 62
    try:
        resynthesis()
        for _ in range(reps):
            t0 = time.perf_counter()
            resynthesis()
            torch.cuda.synchronize()
            single_s.append(time.perf_counter() - t0)
    finally:
        main.FACADE_SPLIT_BYTES = split_bytes
    # the same two flows with copy_out=False: every 'out' a view into the batch's page-locked block (what round 5 timed;
    # since round 6 the default hands out arrays of their own in pageable memory, like decode() — ADVICE r5)
    view_flow, view_rt = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        resynthesis(copy_out=False)
        torch.cuda.synchronize()
        view_flow.append(time.perf_counter() - t0)
    for _ in range(reps):
        t0 = time.perf_counter()
        roundtrip(copy_out=False)
        torch.cuda.synchronize()
        view_rt.append(time.perf_counter() - t0)
    # the thread-per-device driver on the one GPU this process has: devices=[0, 0] — two host threads, two contexts, two
    # streams (world.pool; the same call with devices=[0 .. 7] is how one process drives a node).  Not a scaling figure.
    pool_rt = None
    try:
        def pooled():
            return W.decode_batch(W.encode_batch(fs, xs, f0_method="dio", devices=[0, 0]), devices=[0, 0])
        pooled()
        pooled()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            pooled()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        pool_rt = float(np.median(ts)) * 1e3
    except Exception as ex:  # never costs the block
        pool_rt = "%s: %s" % (type(ex).__name__, ex)
    f, e, d = float(np.median(flow_s)), float(np.median(enc_s)), float(np.median(dec_s))
    return {"resynthesis_flow_ms": f * 1e3, "roundtrip_unmodified_ms": float(np.median(rt_s)) * 1e3,
            "roundtrip_unmodified_pool_devices_0_0_ms": pool_rt,
            "resynthesis_flow_views_ms": float(np.median(view_flow)) * 1e3,
            "roundtrip_unmodified_views_ms": float(np.median(view_rt)) * 1e3,
            "out_arrays": "default: one pageable array per utterance (copied out of the pinned block by the staging threads); "
                          "*_views_ms: copy_out=False, views into the pinned block (round 5's figures)",
            "resynthesis_flow_single_batch_ms": float(np.median(single_s)) * 1e3,
            "parts": 2 if 8 * sum(len(x) for x in xs) >= split_bytes and len(xs) >= 2 else 1,
            "value": frames / f, "unit": "frames/s",
            "x_realtime": len(xs) * len(xs[0]) / fs / f,
            "resynthesis_flow": "World.encode_batch -> scale_pitch(1.5) -> scale_duration(2.0) -> decode_batch on 64 x 10 s, "
                                "NumPy waveforms in, NumPy audio out (2 x 10 s per utterance); dense tensors never read, "
                                "never moved; median of %d" % reps,
            "materialised": {"encode_batch_ms": e * 1e3, "decode_batch_ms": d * 1e3, "value": frames / (e + d),
                             "note": "encode_batch with every dense value read (reference-layout NumPy dicts) + "
                                     "decode_batch of those dicts (everything uploaded again), no modification"}}


def with_transfers_lanes_block(torch, device_index, xs, fs, lanes=4, steps=5):
    """The same host-buffer step with the batch dealt to `lanes` sub-batches on private HIP streams: the D2H of one
    lane's results (PCIe is full duplex and has its own copy engines) runs under the kernels of the next."""
    from world.batch import WorldBatchLanes

    wl = WorldBatchLanes(device_index, lanes=lanes)
    wl.upload(xs, fs)
    parts = wl.split([len(x) for x in xs], lanes)
    x_pins = [torch.from_numpy(np.concatenate(xs[a:b])).pin_memory() for a, b in parts]
    pins = [None] * lanes

    def one(seed):
        for i, (wb, r) in enumerate(zip(wl.lanes, wl.resident)):
            with wb.rt.on_stream():
                batch, x_d, tp_d = r
                x_d.copy_(x_pins[i], non_blocking=True)
                e = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
                yy, _ = wb.decode_device(e, seed=seed, check=False)
                outs = (e.f0, e.vuv, e.spectrogram, e.aperiodicity, yy)
                if pins[i] is None:
                    pins[i] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in outs]
                for p, t in zip(pins[i], outs):
                    p.copy_(t, non_blocking=True)

    one(0)
    wl.synchronize(check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(10 + k)
    wl.synchronize(check=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    frames = wl.total_frames
    return {"ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s",
            "x_realtime": len(xs) * len(xs[0]) / fs / dt, "steps": steps, "lanes": lanes,
            "note": "the with_transfers step dealt to %d lanes (private streams): one lane's D2H under the next "
                    "lane's kernels" % lanes}


def varying_lengths_block(torch, device_index, xs, fs, steps=6):
    """Config 2 with the utterance lengths changing from step to step (two resident batches with different, ragged
    lengths, alternated): every step uploads new per-call metadata (offsets, job tables).  Those uploads are
    stream-ordered copies from pinned staging buffers, so a changing batch costs no device synchronisation; compare
    with `eager_ms_per_step` of the fixed batch."""
    from world.batch import WorldBatch

    wb = WorldBatch(device_index)
    ragged = [x[:len(x) - 800 * (u % 9 + 1)] for u, x in enumerate(xs)]
    res = [wb.upload(xs, fs), wb.upload(ragged, fs)]

    def one(k):
        batch, x_d, tp_d = res[k & 1]
        enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
        return wb.decode_device(enc, seed=k, check=False)

    one(0), one(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(k)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    wb.check("varying_lengths")
    frames = (res[0][0].total_frames + res[1][0].total_frames) / 2
    return {"ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": enq / steps * 1e3, "value": frames / dt,
            "unit": "frames/s", "steps": steps,
            "note": "two resident batches (64 x 10 s and a ragged 64 x 9.55-9.95 s) alternated: per-call tables change "
                    "every step"}


def config1_latency_block(torch, repeats=7):
    """The reference's own benchmark (test/speed.py:13-18 of the reference): ONE World().encode(fs, x,
    f0_method='harvest') — and one decode — on test-mwm.wav (22.05 kHz, 4.64 s, 929 frames) through the drop-in
    facade: host arrays in, host arrays out, every table built, every copy and transpose included."""
    from scipy.io import wavfile

    from world import main
    from world._hip import Runtime

    fs, xi = wavfile.read(os.path.join(ROOT, "tests", "golden", "test-mwm.wav"))
    x = xi / (2 ** 15 - 1)
    W = main.World()
    rt = Runtime.get()

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    dat, first_enc = timed(lambda: W.encode(fs, x, f0_method="harvest"))  # tables of this fs are built here
    _, first_dec = timed(lambda: W.decode(dict(dat)))
    enc_ms, dec_ms = [], []
    for k in range(repeats):
        dat, t = timed(lambda: W.encode(fs, x, f0_method="harvest"))
        enc_ms.append(t)
        _, t = timed(lambda: W.decode(dict(dat)))
        dec_ms.append(t)
    # kernel share of one warm call (HIP events around every launch)
    rt.profile(True)
    dat = W.encode(fs, x, f0_method="harvest")
    k_enc = sum(ms for _, ms in rt.profile_collect())
    W.decode(dict(dat))
    k_dec = sum(ms for _, ms in rt.profile_collect())
    rt.profile(False)
    nf = len(dat["f0"])
    spec_b = dat["spectrogram"].nbytes + dat["aperiodicity"].nbytes + dat["ps spectrogram"].nbytes
    enc, dec = float(np.median(enc_ms)), float(np.median(dec_ms))
    return {"workload": "tests/golden/test-mwm.wav (the reference's test asset): %d samples at %d Hz, %d frames; "
                        "World().encode(fs, x, f0_method='harvest') then World().decode(dat), NumPy in / NumPy out"
                        % (len(x), fs, nf),
            "encode_ms": {"first_call": first_enc, "warm_median": enc, "warm_min": float(np.min(enc_ms)),
                          "warm_all": [round(v, 2) for v in enc_ms], "kernels": k_enc, "host_and_copies": enc - k_enc},
            "decode_ms": {"first_call": first_dec, "warm_median": dec, "warm_min": float(np.min(dec_ms)),
                          "kernels": k_dec, "host_and_copies": dec - k_dec},
            "d2h_bytes_per_encode": int(spec_b + 24 * nf), "h2d_bytes_per_encode": int(x.nbytes * 3),
            "x_realtime_encode": len(x) / fs / (enc / 1e3), "frames_per_s_encode": nf / (enc / 1e3),
            "reference_measured": {"encode_s": 10.07, "decode_s": 0.45,
                                   "source": "BASELINE.md section 2 (authoring container: 8 vCPU Xeon, numba absent)"},
            "note": "first_call: library and context already up (this process ran config 2 before), tables for 22.05 "
                    "kHz not yet built; 'ps spectrogram' (fft x frames complex) is materialised and transposed like "
                    "the reference returns it"}


def _event_ms(torch, fn, reps):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


def feature_heads_block(torch, wl, fs, reps=5):
    """SURVEY §8(f)-2 on the config-2 spectrogram (128 064 frames x 513 bins, resident): encode_lfbank (32 filters),
    encode_mcep (12 coefficients), decode_mcep (12 -> 513), get_context(w=5) — wh_feature_matmul on the FP64 matrix
    cores.  Bytes = input rows + output rows; flops = 2 x rows x k x n (padded sizes are not counted)."""
    from world import features as ft

    wb = wl.lanes[0]
    rt = wb.rt
    batch, x_d, tp_d = wl.resident[0]
    enc = wb.encode_device(batch, x_d, tp_d, fs, f0_method="dio", check=False)
    spec = enc.spectrogram
    f, d = spec.shape
    out = {"frames": int(f), "bins": int(d)}

    def put(name, ms, k, n, in_cols):
        byts = f * (in_cols + n) * 8
        out[name] = {"ms": ms, "GBps": byts / ms / 1e6, "hbm_frac": byts / ms / 1e6 / HBM_PEAK_GBS,
                     "TFLOPs": 2.0 * f * k * n / ms / 1e9, "frames_per_s": f / ms * 1e3}

    ft.lfbank_device(rt, spec), ft.mcep_device(rt, spec)
    ms, lf = _event_ms(torch, lambda: ft.lfbank_device(rt, spec), reps)
    put("encode_lfbank_32", ms, d, 32, d)
    ms, mc = _event_ms(torch, lambda: ft.mcep_device(rt, spec), reps)
    put("encode_mcep_12", ms, d, 12, d)
    ft.imcep_device(rt, mc, 2 * (d - 1))
    ms, _ = _event_ms(torch, lambda: ft.imcep_device(rt, mc, 2 * (d - 1)), reps)
    put("decode_mcep_12", ms, 12, d, 12)
    ft.context_device(rt, lf, 5)
    ms, _ = _event_ms(torch, lambda: ft.context_device(rt, lf, 5), reps)
    out["get_context_w5"] = {"ms": ms, "GBps": f * 32 * 12 * 8 / ms / 1e6, "hbm_frac": f * 32 * 12 * 8 / ms / 1e6 / HBM_PEAK_GBS}
    wb.check("feature_heads")
    return out


def swipe_block(torch, wl, fs, reps=3):
    """SURVEY §8(f)-3 on the config-2 batch (64 x 10 s, resident): f0_method='swipe' — STFTs at every window size, the
    two dense products per window on the FP64 matrix cores, strength interpolation and the parabolic pick."""
    from world.swipe import swipe_device, swipe_tables

    wb = wl.lanes[0]
    rt = wb.rt
    batch, x_d, tp_d = wl.resident[0]
    swipe_device(rt, batch, x_d, fs, (71, 800), 0.005, 0.3)
    t0 = time.perf_counter()
    swipe_device(rt, batch, x_d, fs, (71, 800), 0.005, 0.3)
    host_ms = (time.perf_counter() - t0) * 1e3
    ms, _ = _event_ms(torch, lambda: swipe_device(rt, batch, x_d, fs, (71, 800), 0.005, 0.3), reps)
    rt.profile(True)
    swipe_device(rt, batch, x_d, fs, (71, 800), 0.005, 0.3)
    agg = {}
    for name, t in rt.profile_collect():
        agg[name] = agg.get(name, 0.0) + t
    rt.profile(False)
    wb.check("swipe")
    tb = swipe_tables(int(fs), 71.0, 800.0)
    frames = batch.total_frames
    # dense-product flops: per window size, segments x (bins x n_erb + n_erb x n_c) x 2
    n = batch.total_samples / batch.n_utt
    flops = 0.0
    for w in tb["windows"]:
        seg = batch.n_utt * (n + w["ws"] / 2 + w["hop"] + w["ws"] / 2) / w["hop"]
        flops += 2.0 * seg * ((w["ws"] // 2 + 1) * tb["n_erb"] + tb["n_erb"] * w["n_c"])
    return {"ms": ms, "frames_per_s": frames / ms * 1e3, "x_realtime": batch.total_samples / fs / (ms / 1e3),
            "host_enqueue_ms": host_ms, "matmul_TFLOPs": flops / (agg.get("feature_matmul_kernel", ms)) / 1e9,
            "matmul_flops": flops, "window_sizes": [w["ws"] for w in tb["windows"]], "candidates": len(tb["pc"]),
            "kernel_ms": {k: round(v, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}}


def time_pipelines(torch, pipes, steps):
    """`pipes`: [(step function, HIP stream or None)] — independent pipelines over copies of one batch.  Times `steps`
    eager steps on pipeline 0 (host enqueue time and wall time per step), then, replayed from one hipGraph per pipeline,
    `steps` steps dealt round-robin to all pipelines and `steps` steps on pipeline 0 alone.  Returns a dict of seconds per
    step: enqueue, eager, pipelined (all in flight), one (one in flight), and graph (bool)."""
    fn0 = pipes[0][0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fn0(1 + k)
    enq = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    graphs = [try_capture(torch, (lambda fn=fn: fn(1000)), stream=st) for fn, st in pipes]
    out = {"enqueue": enq, "eager": eager, "pipelined": eager, "one": eager, "graph": False}
    if all(g is not None for g in graphs):
        def run(active):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(steps):
                d = k % active
                if pipes[d][1] is not None:
                    with torch.cuda.stream(pipes[d][1]):
                        graphs[d].replay()
                else:
                    graphs[d].replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps
        out["pipelined"] = run(len(pipes))
        out["one"] = run(1) if len(pipes) > 1 else out["pipelined"]
        out["graph"] = True
    del graphs
    return out


def other_configs_block(torch, device_index, xs16, xs48, depth=2):
    """BASELINE configs 3, 4 and 5 at their single-GPU sizes, timed by this process like the headline (hipGraph replays,
    `depth` whole steps in flight; the one-in-flight figure, the eager figure, the host's enqueue time per step and the sum
    of the kernels' HIP-event durations beside it, so that a gap between the step and its kernels is visible in the line):
    3 = Harvest only on 256 x 10 s (the size BASELINE.json states it on); 4 = Harvest + CheapTrick + D4C-Requiem encode +
    Requiem decode on 64 x 10 s; 5 = 16 x 60 s at 48 kHz, Harvest encode, scale_pitch(1.5), scale_duration(2.0), decode."""
    import types

    from world import _hip
    from world.batch import WorldBatchLanes

    out = {}
    # config 3 is stated on 256 x 10 s (BASELINE.json configs[2]): the 64 distinct utterances four times over
    xs16_256 = None if xs16 is None else [xs16[i % len(xs16)] for i in range(256)]
    for cfg, xs, fs, steps in ((3, xs16_256, 16000, 6), (4, xs16, 16000, 6), (5, xs48, 48000, 4)):
        if xs is None:
            out["config%d" % cfg] = {"error": "inputs unavailable"}
            continue
        wls = [WorldBatchLanes(device_index, lanes=1, first_lane=(d + 1) if depth > 1 else None) for d in range(depth)]
        pipes = []
        for w_ in wls:
            w_.upload(xs, fs)
            pipes.append((make_step(types.SimpleNamespace(config=cfg, no_stagger=True), w_, fs), w_.lanes[0].rt.own_stream))
        for fn, _ in pipes:
            fn(0)
        torch.cuda.synchronize()
        t = time_pipelines(torch, pipes, steps)
        dt = t["pipelined"]
        wl = wls[0]
        rt = wl.lanes[0].rt
        rt.profile(True)
        pipes[0][0](99)
        agg = {}
        for name, ms in rt.profile_collect():
            agg[name] = agg.get(name, 0.0) + ms
        rt.profile(False)
        for w_ in wls:
            w_.lanes[0].check("other_configs %d" % cfg)
        frames = wl.total_frames
        out["config%d" % cfg] = {"utterances": len(xs), "distinct_utterances": min(len(xs), 64 if fs == 16000 else 16),
                                 "seconds": len(xs[0]) / fs, "fs": fs, "steps": steps,
                                 "ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s",
                                 "x_realtime": len(xs) * len(xs[0]) / fs / dt, "steps_in_flight": depth,
                                 "ms_per_step_one_in_flight": t["one"] * 1e3,
                                 "graph": t["graph"], "eager_ms_per_step": t["eager"] * 1e3,
                                 "host_enqueue_ms_per_step": t["enqueue"] * 1e3, "kernel_ms_sum": sum(agg.values()),
                                 "kernel_ms": {k: round(v, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:6]}}
        del wls, wl, pipes
        _hip.Runtime.trim_all()
    return out


def north_star_block(torch, device_index, xs_distinct, fs, args, steps=6):
    """BASELINE.json north_star on ONE GPU: 1024 x 10 s at 16 kHz, encode(harvest, is_requiem=True) + Requiem decode.
    Timed like the headline: `steps` (>= 5) replays of hipGraphs of one step between synchronisations, dealt to
    --in-flight pipelines (each with its own resident copy of the batch, context, arena and stream), the one-in-flight and
    eager figures beside it; per-kernel HIP-event durations from one un-captured step, and a `roofline` block of its own
    for the dominant kernel with the HBM traffic MEASURED in this run (two rocprofv3 --pmc child runs of config 4 at this
    size)."""
    from world.batch import WorldBatch

    n = args.north_star_utts
    xs = [xs_distinct[i % len(xs_distinct)] for i in range(n)]
    from world import _hip
    _hip.Runtime.trim_all()  # the arenas of the earlier blocks (they only grow) go back to the device first
    # Steps in flight: the workspace of one pipeline at 1024 utterances is ~96 GB (47 GB of crossing lists sized with 3 x
    # head-room, 17 GB of refined candidates ...; round 5: 105 GB, with the raw-candidate map).  A second pipeline — its own
    # resident copy of the batch, context, arena, stream — is added when --in-flight asks for it AND, measured after the
    # first one has run a step, the device still has 1.4 x that much free (the graphs' private pools come on top); the
    # serial head of a step is 1-2 % of it at this size, so it buys ~1 % (194.4 -> 192.0 ms, tools/ns_mem.py).
    want = max(1, getattr(args, "in_flight", 1))
    free0, _ = torch.cuda.mem_get_info()
    wbs = [WorldBatch(device_index, lane=1 if want > 1 else 0)]
    res = [wbs[0].upload(xs, fs)]

    def make_one(w, r):
        def one():
            enc = w.encode_device(r[0], r[1], r[2], fs, f0_method="harvest", is_requiem=True, check=False)
            return w.decode_device(enc, check=False)  # device-generated seed tables
        return one

    ones = [make_one(wbs[0], res[0])]
    ones[0]()
    torch.cuda.synchronize()
    wbs[0].check("north_star warm-up")
    free1, _ = torch.cuda.mem_get_info()
    pipeline_bytes = free0 - free1
    if want > 1 and free1 >= 1.4 * pipeline_bytes:
        wbs.append(WorldBatch(device_index, lane=2))
        res.append(wbs[1].upload(xs, fs))
        ones.append(make_one(wbs[1], res[1]))
        ones[1]()
        torch.cuda.synchronize()
        wbs[1].check("north_star warm-up")
    depth = len(wbs)
    one = ones[0]
    wb, (batch, x_d, tp_d) = wbs[0], res[0]
    t = time_pipelines(torch, [((lambda k, fn=fn: fn()), w.rt.own_stream) for fn, w in zip(ones, wbs)], steps)
    enq, eager, dt, dt_one = t["enqueue"], t["eager"], t["pipelined"], t["one"]
    wb.rt.profile(True)  # per-kernel durations from one more step (the event pairs stay out of the timed steps)
    one()
    agg = {}
    for name, ms in wb.rt.profile_collect():
        a = agg.setdefault(name, [0.0, 0])
        a[0] += ms
        a[1] += 1
    wb.rt.profile(False)
    for w in wbs:
        w.check("north_star")
    frames = batch.total_frames
    dom = max(agg.items(), key=lambda kv: kv[1][0])[0]
    per_k, path_b = algo_bytes_per_frame(fs, 1024, requiem=True)
    dom_ms = agg[dom][0] / agg[dom][1]
    # Requiem path: 9584 B/frame encode+decode; Harvest kernels are priced on the F0-only 664 B/frame
    per_frame = per_k.get(dom, 664)
    frames_per_launch = frames / agg[dom][1]
    achieved = per_frame * frames_per_launch / (dom_ms / 1e3) / 1e9
    traffic, traffic_all, traffic_src = None, None, {"source": "not measured (--no-pmc)"}
    if not args.no_pmc:
        del x_d, tp_d, batch, res, ones, one  # the child runs need the HBM this block holds no longer
        _hip.Runtime.trim_all()
        measured, how = measure_pmc_traffic(args, timeout_s=420, config=4, utts=n, seconds=len(xs[0]) / fs, steps=1)
        if measured and dom in measured:
            traffic, traffic_all, traffic_src = measured[dom], measured, {"source": how}
        else:
            traffic_src = {"source": "measurement unavailable", "reason": how if not measured else "kernel not in the trace"}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic": None if traffic is None else traffic / (per_frame * frames_per_launch),
                "traffic_per_kernel_MB": None if traffic_all is None else
                {k: round(v / 1e6, 1) for k, v in sorted(traffic_all.items(), key=lambda kv: -kv[1])[:8]},
                "algorithmic_bytes_per_launch": per_frame * frames_per_launch, "frames_per_launch": frames_per_launch,
                "avg_launch_ms": dom_ms, "timing": "HIP events around every launch, one un-captured step",
                "path_algorithmic_GBps": path_b * frames / dt / 1e9}
    kms = {k: v[0] / v[1] for k, v in agg.items()}
    live = fp64_view(dom, dom_ms) if traffic_all is not None else None
    if live:
        roofline.update({k: live[k] for k in ("fp64_TFLOPs", "fp64_peak_frac", "valu_busy", "fp64_inst_share")})
        roofline["fp64_per_kernel"] = {k: {kk: round(v[kk], 4) for kk in ("fp64_TFLOPs", "fp64_peak_frac", "valu_busy", "fp64_inst_share")}
                                       for k, v in ((k, fp64_view(k, kms[k])) for k in sorted(kms, key=lambda q: -agg[q][0])[:6]) if v}
    if traffic_all is not None:  # the Harvest front chain on its own (91 GB in round 5)
        front = [k for k in ("band_events_kernel", "hv_rawdet_kernel", "hv_raw_kernel", "hv_detect_kernel") if k in traffic_all]
        roofline["harvest_front_chain_GB"] = {"kernels": front, "total": round(sum(traffic_all[k] for k in front) / 1e9, 2)}
    return {"workload": "%d x %.0f s synthetic 16 kHz utterances (%d distinct) on 1 GPU: Harvest+CheapTrick+"
                        "D4C-Requiem encode + Requiem decode, HBM-resident" % (n, len(xs[0]) / fs, min(n, len(xs_distinct))),
            "distinct_utterances": min(n, len(xs_distinct)), "host_enqueue_ms_per_step": enq * 1e3,
            "graph": t["graph"], "eager_ms_per_step": eager * 1e3, "steps_in_flight": depth,
            "pipeline_device_GB": round(pipeline_bytes / 2 ** 30, 1),
            "ms_per_step_one_in_flight": dt_one * 1e3,
            "ms_per_step": dt * 1e3, "value": frames / dt, "unit": "frames/s", "x_realtime": n * len(xs[0]) / fs / dt,
            "target_x_realtime": 500, "steps": steps, "frames_per_step": frames,
            "roofline": roofline, "cpu_baseline": getattr(args, "north_star_cpu", None),
            "kernel_ms": {k: round(v[0] / v[1], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]}}


if __name__ == "__main__":
    main()
