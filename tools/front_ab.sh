# A/B of the fused Harvest front end on one box: tests first, then config 3 (64 and 1024 utterances) with
# WH_HV_FRONT=0 (unfused chain) and 1.  usage (through gpurun): bash tools/front_ab.sh <outdir under gpurun_out>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_harvest.py tests/test_hip_longform.py tests/test_hip_differential.py tests/test_hip_determinism.py tests/test_hip_edge_cases.py tests/test_hip_parameters.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for f in 0 1; do
  for n in 64 1024; do
    WH_HV_FRONT=$f python bench.py --config 3 --utts $n --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/b3_f${f}_n$n.json 2> $O/b3_f${f}_n$n.err
    python - "$O/b3_f${f}_n$n.json" "front=$f n=$n" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-18s %8.3f ms/step  %s" % (sys.argv[2], d["ms_per_step"], {k: v for k, v in list(d["kernel_ms"].items())[:8]}))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
