#!/bin/bash
# Eight ranks of the scaling bench time-sharing ONE GPU (WH_BENCH_SHARE_GPU=1, gloo): the N = 8 code path — sharding by
# shard_ranges, per-rank input generation, barrier, max / sum reductions, per_rank_ms — run end to end at the widths the
# targets are quoted on.  Not a scaling measurement.  tools/rehearsal_n8.sh <outdir>
O=${1:-gpurun_out/rehearsal_n8}
mkdir -p $O
TR="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
# north star: 1024 x 10 s of config 4, strong scaling -> 128 utterances per rank
WH_BENCH_SHARE_GPU=1 timeout 900 $TR --nproc-per-node 8 --master-port 29531 bench.py --gpus 8 --config 4 --utts 1024 --scaling strong --steps 3 --warmup 1 --in-flight 1 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_share_gpu_n8_strong_cfg4_1024.log 2>&1
# config 5 (128 x 60 s at 48 kHz over 8 GPUs = 16 per rank) at what one GPU's memory holds for eight ranks: 4 per rank
WH_BENCH_SHARE_GPU=1 timeout 900 $TR --nproc-per-node 8 --master-port 29532 bench.py --gpus 8 --config 5 --utts 32 --scaling strong --steps 2 --warmup 1 --in-flight 1 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_share_gpu_n8_strong_cfg5_32.log 2>&1
# weak scaling, the metric's config: 8 x 64 utterances
WH_BENCH_SHARE_GPU=1 timeout 900 $TR --nproc-per-node 8 --master-port 29533 bench.py --gpus 8 --steps 4 --warmup 1 --in-flight 1 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_share_gpu_n8_weak.log 2>&1
for f in $O/rehearsal_share_gpu_n8_*.log; do echo "== $f"; grep '^{' $f | tail -1 | cut -c1-700; done
