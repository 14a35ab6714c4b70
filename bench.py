#!/usr/bin/env python3
"""Benchmark of the MI355X WORLD hot path (BASELINE.json metric: analysis+synthesis frames/s and xRT,
16 kHz / 5 ms hop).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling): BASELINE config 2 per GPU — 64 x 10 s synthetic 16 kHz utterances,
encode (DIO + StoneMask + CheapTrick + D4C) + decode (pulse-wise synthesis), inputs resident in HBM before
the timed region, outputs left in HBM.  A "step" is one full encode+decode pass over the rank's batch.
Utterances are independent: ranks shard them with NO collective on the data path; the only communication
is the barrier / max-reduce of the timing.

Rank 0 prints ONE JSON line (see README / DESIGN.md §Measurement for the field definitions).
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
def algo_bytes_per_frame(fs, fft_size, out_hop_scale=1.0):
    hop = int(fs * 5 // 1000) * 8
    kb = (fft_size // 2 + 1) * 8
    per_kernel = {
        "cheaptrick_kernel": hop + 24 + kb,
        "d4c_kernel": hop + 24 + kb,
        "love_train_kernel": hop + 24 + 4,
        "response_kernel": 24 + kb + kb + int(hop * out_hop_scale),
        "stonemask_kernel": hop + 24 + 8,
        # Harvest kernels: the F0-only path reads the hop and writes f0/vuv/tp (664 B/frame)
        "hv_refine_kernel": hop + 24,
        "band_events_kernel": hop + 24,
    }
    path = hop + int(hop * out_hop_scale) + 48 + 4 * kb  # whole encode+decode path: 17744 B at 16 kHz
    return per_kernel, path


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=None,
                    help="utterances per GPU (default: 64 = BASELINE config 2; 16 for config 5)")
    ap.add_argument("--seconds", type=float, default=None, help="utterance length (default 10 s; 60 s for config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=12, help="utterances of the batch timed through the CPU oracle")
    ap.add_argument("--lanes", type=int, default=1,
                    help="independent sub-batches in flight per GPU, each on its own HIP stream (world.batch."
                         "WorldBatchLanes).  2 staggered lanes measure ~6%% faster on config 2, but the per-kernel "
                         "HIP-event durations then include the other lane's overlap, so the default keeps one lane and "
                         "clean per-kernel attribution")
    ap.add_argument("--no-stagger", action="store_true", help="start all lanes together (ablation)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 = DIO path encode+decode (the metric's config, default); "
                         "3 = Harvest F0 only; 4 = Harvest + CheapTrick + D4C-Requiem encode + Requiem decode; "
                         "5 = 48 kHz long-form: encode, scale_pitch(1.5), scale_duration(2.0), decode")
    args = ap.parse_args()
    if args.utts is None:
        args.utts = 16 if args.config == 5 else UTT_PER_GPU
    if args.seconds is None:
        args.seconds = 60.0 if args.config == 5 else SECONDS
    return args


def pmc_traffic(kernel, lanes):
    """HBM bytes per launch of `kernel` measured with rocprofv3 PMC passes on this workload (committed under
    profiles/; bench.py itself cannot run the profiler around its own timed region).  The file holds bytes per
    pass over the whole 64-utterance batch; one launch covers 1/lanes of it.  None if unavailable."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic_cfg2_latest.txt")
    try:
        for line in open(path):
            parts = line.split()
            if parts and parts[0] == kernel:
                return float(parts[3]) * 1e6 / lanes, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def pmc_fp64_flops(kernel, lanes):
    """FP64 floating-point operations per launch of `kernel` (2*FMA + ADD + MUL + TRANS instructions x 64 lanes) from the
    committed rocprofv3 SQ-counter digest of this workload (profiles/sq_counters_cfg2_latest.txt).  None if unavailable."""
    path = os.path.join(ROOT, "profiles", "sq_counters_cfg2_latest.txt")
    try:
        for line in open(path):
            parts = line.split()
            if parts and parts[0] == kernel:
                return float(parts[1]) * 1e-6 * float(parts[-1]) * 1e12 / lanes, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def cpu_baseline(xs, n_utts):
    """The NumPy oracle (a 'port' of the reference path) on a bounded sample of the same workload."""
    from oracle import api as oapi

    n_utts = min(n_utts, len(xs))
    frames = 0
    t0 = time.perf_counter()
    for u in range(n_utts):
        dat = oapi.encode_np(FS, xs[u], f0_method="dio")
        np.random.seed(u)
        oapi.decode_np(dat)
        frames += len(dat["f0"])
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d utterances (%.0f s each), encode(dio)+decode through oracle/ (NumPy), %.1f s wall"
                      % (n_utts, len(xs), len(xs[0]) / FS, dt),
            "x_realtime": n_utts * len(xs[0]) / FS / dt}


def main():
    global FS
    args = parse()
    if args.config == 5:
        FS = 48000
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run the RCCL path is exercised even for 1 rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from world._synthetic import synth_utterance
    from world.batch import WorldBatchLanes

    # args.lanes independent sub-batches per GPU, each on its own HIP stream and library context; one "step" is
    # still one pass of the hot path over the whole per-GPU batch
    wl = WorldBatchLanes(local_rank, lanes=max(1, min(args.lanes, args.utts)))
    rts = [wb.rt for wb in wl.lanes]
    first = rank * args.utts
    xs = [synth_utterance(first + i, FS, args.seconds) for i in range(args.utts)]
    wl.upload(xs, FS)  # inputs resident in HBM before the timed region
    frames_per_step = wl.total_frames

    if args.config == 2:
        def step(seed):
            encs = wl.encode_device(FS, stagger=not args.no_stagger, f0_method="dio")
            return wl.decode_device(encs, seed=seed)
    elif args.config == 3:
        from world.harvest import harvest_device

        def step(seed):
            out = []
            for wb, r in zip(wl.lanes, wl.resident):
                with wb.rt.on_stream():
                    out.append(harvest_device(wb.rt, r[0], r[1], r[2], FS))
            return out
    elif args.config == 5:
        def step(seed):
            encs = wl.encode_device(FS, stagger=not args.no_stagger, f0_method="dio")
            for e in encs:
                with e.rt.on_stream():
                    e.scale_pitch(1.5).scale_duration(2.0)
            return wl.decode_device(encs, seed=seed)
    else:
        from world.get_seeds_signals import get_seeds_signals
        import random
        random.seed(0)
        np.random.seed(0)
        seeds = get_seeds_signals(FS)

        def step(seed):
            encs = wl.encode_device(FS, stagger=not args.no_stagger, f0_method="harvest", is_requiem=True)
            return wl.decode_device(encs, seeds=seeds)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    fence()
    for rt in rts:
        rt.profile(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(1000 + k)
    host_enqueue = time.perf_counter() - t0  # the launches are asynchronous: host time to enqueue all K steps
    fence()
    elapsed = time.perf_counter() - t0
    records, flags = [], [0] * 16
    for rt in rts:
        records += rt.profile_collect()
        rt.profile(False)
        flags = [a | b for a, b in zip(flags, rt.take_flags())]

    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        audio_s = args.utts * args.seconds * world * args.steps
        value = total_frames / elapsed
        # per-kernel launch durations from the HIP-event records of the timed region (rank 0, each lane's stream)
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
            ALGO_BYTES_PER_FRAME, PATH_BYTES_PER_FRAME = algo_bytes_per_frame(FS, default_fft_size(FS),
                                                                              2.0 if args.config == 5 else 1.0)
            per_frame = ALGO_BYTES_PER_FRAME.get(dominant, PATH_BYTES_PER_FRAME)
            avg_s = kernel_ms[dominant] / 1e3
            # every lane launches the kernel once per step on its share of the frames
            frames_per_launch = frames_per_step * args.steps / agg[dominant][1]
            achieved = per_frame * frames_per_launch / avg_s / 1e9
            traffic, traffic_src = (pmc_traffic(dominant, len(rts)) if args.config == 2 and args.utts == UTT_PER_GPU
                                    else (None, None))
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": per_frame * frames_per_launch,
                        "frames_per_launch": frames_per_launch,
                        "avg_launch_ms": kernel_ms[dominant],
                        "path_algorithmic_GBps": PATH_BYTES_PER_FRAME * frames_per_step * args.steps / elapsed / 1e9}
            # second view of the same kernel: the path is FP64-compute/latency-bound, not HBM-bound
            flops, flops_src = (pmc_fp64_flops(dominant, len(rts)) if args.config == 2 and args.utts == UTT_PER_GPU
                                else (None, None))
            if flops:
                roofline["fp64_vector"] = {"achieved": flops / avg_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": flops / avg_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                           "flops_per_launch": flops, "flops_source": flops_src}
        out = {
            "metric": "analysis+synthesis frames/sec (and xRT), %d kHz / 5 ms hop" % (FS // 1000),
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "x_realtime": audio_s / elapsed,
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
            "config": {"workload": {2: "BASELINE config 2 per GPU: %d x %.0f s synthetic 16 kHz utterances, "
                                       "DIO+StoneMask+CheapTrick+D4C encode + pulse-wise synthesis decode, HBM-resident",
                                    3: "BASELINE config 3 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest F0 only",
                                    4: "BASELINE config 4 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest+CheapTrick+"
                                       "D4C-Requiem encode + Requiem decode",
                                    5: "BASELINE config 5 per GPU: %d x %.0f s synthetic 48 kHz utterances, DIO+StoneMask+"
                                       "CheapTrick+D4C encode, scale_pitch(1.5), scale_duration(2.0), pulse-wise decode"
                                    }[args.config] % (args.utts, args.seconds),
                       "utterances_per_gpu": args.utts, "fs": FS, "frame_period_ms": 5,
                       "frames_per_step_per_gpu": frames_per_step, "sharding": "utterances, no collective",
                       "lanes_per_gpu": len(rts)},
            "roofline": roofline,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])},
            "device_flags": flags,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            out["cpu_baseline"] = cpu_baseline(xs, args.cpu_utts)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
