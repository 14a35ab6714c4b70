#!/bin/bash
# Final-state record of a round (through gpurun): tools/final_record.sh <tag>  ->  gpurun_out/<tag>/
#   the whole GPU suite, smoke(), the default bench line, configs 3 / 4 / 5 on their own, the two-rank rehearsals of the
#   scaling bench on one GPU (WH_BENCH_SHARE_GPU=1: weak, and config 4 strong), the torchrun N = 1 run under RCCL, and the
#   rocprofv3 passes of configs 2, 3 and 4 (one step in flight, so that a kernel's counters are its own).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python bench.py --config 3 --steps 10 --warmup 2 --no-pmc > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 3 --utts 256 --steps 6 --warmup 1 --no-pmc > $O/bench_cfg3_256.json 2> $O/bench_cfg3_256.err
python bench.py --config 4 --steps 10 --warmup 2 --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config 5 --steps 4 --warmup 1 --no-pmc > $O/bench_cfg5.json 2> $O/bench_cfg5.err
TR="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
$TR --nproc-per-node 1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_torchrun_n1.log 2>&1
WH_BENCH_SHARE_GPU=1 $TR --nproc-per-node 2 --master-port 29518 bench.py --gpus 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_share_gpu_n2_weak.log 2>&1
WH_BENCH_SHARE_GPU=1 $TR --nproc-per-node 2 --master-port 29519 bench.py --gpus 2 --config 4 --utts 128 --scaling strong --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_share_gpu_n2_strong_cfg4.log 2>&1
python - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")) + sorted(glob.glob(sys.argv[1] + "/rehearsal_*.log")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "ms/step %.3f (one in flight %s)" % (d["ms_per_step"], d.get("ms_per_step_one_in_flight")), "n_gpus", d["n_gpus"], "value %.4g" % d["value"],
              "xRT %.0f" % d["x_realtime"], "graph", d["graph"], "per_rank_ms", d.get("per_rank_ms"), list(d["kernel_ms"].items())[:4])
    except Exception as e:
        print(f, "ERR", e)
PY
for c in 2 3 4; do timeout 900 tools/profile_suite.sh $c $1/prof_cfg$c --in-flight 1 > $O/prof$c.log 2>&1; echo "profile cfg$c rc=$?"; done
