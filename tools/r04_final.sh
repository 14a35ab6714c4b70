# Final-state record of round 4 (through gpurun): the whole GPU suite, the default bench line, configs 3/4/5 on their own,
# config 3 at the stated 256 utterances, the torchrun N=1 rehearsal.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config 3 --steps 5 --warmup 2 --no-pmc > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 3 --utts 256 --steps 3 --warmup 1 --no-pmc > $O/bench_cfg3_256.json 2> $O/bench_cfg3_256.err
python bench.py --config 4 --steps 5 --warmup 2 --no-pmc > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config 5 --steps 2 --warmup 1 --no-pmc > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-pmc > $O/rehearsal_torchrun_n1.log 2>&1
python - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")) + [sys.argv[1] + "/rehearsal_torchrun_n1.log"]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "value %.4g" % d["value"], "xRT %.0f" % d["x_realtime"], "graph", d["graph"], list(d["kernel_ms"].items())[:4])
    except Exception as e:
        print(f, "ERR", e)
PY
