#!/bin/bash
# Final-state record of a round (through gpurun): tools/final_record.sh <tag>  ->  gpurun_out/<tag>/
#   the whole GPU suite, smoke(), the default bench line, configs 3 / 4 / 5 on their own, the two-rank rehearsals of the
#   scaling bench on one GPU (WH_BENCH_SHARE_GPU=1: weak, and config 4 strong), the torchrun N = 1 run under RCCL, and the
#   rocprofv3 passes of configs 2 - 5 and of config 4 at the north-star size (one step in flight, so that a kernel's
#   counters are its own), digested on the box into $O/digests/ (what gets copied to profiles/), and the eight-rank
#   rehearsal (tools/rehearsal_n8.sh).
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
bash tools/rehearsal_n8.sh $O > $O/rehearsal_n8_summary.txt 2>&1; tail -6 $O/rehearsal_n8_summary.txt | cut -c1-300
mkdir -p $O/digests
digest() {  # digest <config> <dir> <frames per launch> <label> [pmc_traffic_summary key=value ...]
  local c=$1 d=$2 fr=$3 label=$4; shift 4
  cp $d/trace/t_kernel_stats.csv $O/digests/cfg${c}_kernel_stats.csv 2>/dev/null
  python tools/pmc_traffic_summary.py $d/fetch/f_counter_collection.csv $d/write/w_counter_collection.csv $fr "$label" "$@" > $O/digests/cfg${c}_hbm_traffic_pmc.txt 2>> $O/digest.err
  python tools/sq_counters_summary.py $d/sqa/a_counter_collection.csv $d/sqb/b_counter_collection.csv $d/trace/t_kernel_trace.csv "config $c" > $O/digests/cfg${c}_sq_counters.txt 2>> $O/digest.err
  rm -rf $d/fetch $d/write $d/sqa $d/sqb $d/trace/t_kernel_trace.csv
}
for c in 2 3 4 5; do timeout 900 tools/profile_suite.sh $c $1/prof_cfg$c --in-flight 1 > $O/prof$c.log 2>&1; echo "profile cfg$c rc=$?"; done
digest 2 $O/prof_cfg2 128064 "config 2 (64 x 10 s)"
digest 3 $O/prof_cfg3 128064 "config 3 (64 x 10 s, Harvest only)"
digest 4 $O/prof_cfg4 128064 "config 4 (64 x 10 s, Requiem)" fs=16000 fft=1024 requiem=1
digest 5 $O/prof_cfg5 192016 "config 5 (16 x 60 s, 48 kHz)" fs=48000 fft=2048 out_hop_scale=2
timeout 1200 tools/profile_suite.sh 4 $1/prof_northstar --utts 1024 --in-flight 1 > $O/prof_ns.log 2>&1; echo "profile north star rc=$?"
digest northstar_1024 $O/prof_northstar 2049024 "config 4 at the north-star size (1024 x 10 s, Requiem)" fs=16000 fft=1024 requiem=1
ls $O/digests
