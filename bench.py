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

# Algorithmic (compulsory) HBM bytes per 5 ms frame of each dominant-kernel candidate at 16 kHz,
# fft 1024, float64 API dtypes — SURVEY.md §8(d) components: x hop 640 B, f0+vuv+tp 24 B,
# spectrogram 4104 B, aperiodicity 4104 B, output hop 640 B.
ALGO_BYTES_PER_FRAME = {
    "cheaptrick_kernel": 640 + 24 + 4104,
    "d4c_kernel": 640 + 24 + 4104,
    "love_train_kernel": 640 + 24 + 4,
    "response_kernel": 24 + 4104 + 4104 + 640,
    "stonemask_kernel": 640 + 24 + 8,
    # Harvest kernels: the F0-only path reads the hop and writes f0/vuv/tp (664 B/frame)
    "hv_refine_kernel": 640 + 24,
    "band_events_kernel": 640 + 24,
}
PATH_BYTES_PER_FRAME = 17744  # whole encode+decode path, SURVEY §8(d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=UTT_PER_GPU, help="utterances per GPU (default: BASELINE config 2)")
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=12, help="utterances of the batch timed through the CPU oracle")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4],
                    help="BASELINE.json config: 2 = DIO path encode+decode (the metric's config, default); "
                         "3 = Harvest F0 only; 4 = Harvest + CheapTrick + D4C-Requiem encode + Requiem decode")
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` measured with rocprofv3 PMC passes on this workload (committed under
    profiles/; bench.py itself cannot run the profiler around its own timed region).  None if unavailable."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic_cfg2_latest.txt")
    try:
        for line in open(path):
            parts = line.split()
            if parts and parts[0] == kernel:
                return float(parts[3]) * 1e6, os.path.relpath(path, ROOT)
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
    args = parse()
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
    from world.batch import WorldBatch

    wb = WorldBatch(local_rank)
    rt = wb.rt
    first = rank * args.utts
    xs = [synth_utterance(first + i, FS, args.seconds) for i in range(args.utts)]
    batch, x_d, tp_d = wb.upload(xs, FS)  # inputs resident in HBM before the timed region
    frames_per_step = batch.total_frames

    if args.config == 2:
        def step(seed):
            enc = wb.encode_device(batch, x_d, tp_d, FS, f0_method="dio")
            y, _ = wb.decode_device(enc, seed=seed)
            return y
    elif args.config == 3:
        from world.harvest import harvest_device

        def step(seed):
            return harvest_device(rt, batch, x_d, tp_d, FS)
    else:
        from world.get_seeds_signals import get_seeds_signals
        import random
        random.seed(0)
        np.random.seed(0)
        seeds = get_seeds_signals(FS)

        def step(seed):
            enc = wb.encode_device(batch, x_d, tp_d, FS, f0_method="harvest", is_requiem=True)
            y, _ = wb.decode_device(enc, seeds=seeds)
            return y

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    fence()
    rt.profile(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(1000 + k)
    fence()
    elapsed = time.perf_counter() - t0
    records = rt.profile_collect()
    rt.profile(False)
    flags = rt.take_flags()

    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        audio_s = args.utts * args.seconds * world * args.steps
        value = total_frames / elapsed
        # per-kernel totals from the HIP-event records of the timed region (rank 0's stream)
        agg = {}
        for name, ms in records:
            a = agg.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        kernel_ms = {k: v[0] / v[1] for k, v in agg.items()}
        dominant = max(agg.items(), key=lambda kv: kv[1][0])[0] if agg else None
        roofline = None
        if dominant:
            per_frame = ALGO_BYTES_PER_FRAME.get(dominant, PATH_BYTES_PER_FRAME)
            avg_s = kernel_ms[dominant] / 1e3
            achieved = per_frame * frames_per_step / avg_s / 1e9
            traffic, traffic_src = pmc_traffic(dominant) if args.config == 2 and args.utts == UTT_PER_GPU else (None, None)
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": per_frame * frames_per_step,
                        "avg_launch_ms": kernel_ms[dominant],
                        "path_algorithmic_GBps": PATH_BYTES_PER_FRAME * frames_per_step * args.steps / elapsed / 1e9}
        out = {
            "metric": "analysis+synthesis frames/sec (and xRT), 16 kHz / 5 ms hop",
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
            "config": {"workload": {2: "BASELINE config 2 per GPU: %d x %.0f s synthetic 16 kHz utterances, "
                                       "DIO+StoneMask+CheapTrick+D4C encode + pulse-wise synthesis decode, HBM-resident",
                                    3: "BASELINE config 3 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest F0 only",
                                    4: "BASELINE config 4 per GPU: %d x %.0f s synthetic 16 kHz utterances, Harvest+CheapTrick+"
                                       "D4C-Requiem encode + Requiem decode"}[args.config] % (args.utts, args.seconds),
                       "utterances_per_gpu": args.utts, "fs": FS, "frame_period_ms": 5,
                       "frames_per_step_per_gpu": frames_per_step, "sharding": "utterances, no collective"},
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
