cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_batch.py -m gpu -q 2>&1 | tail -2; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --north-star-utts 64 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3w/bench_cfg2.json'))
print(d['ms_per_step'])
for k in ('with_transfers','with_transfers_pipelined','varying_lengths'):
    print(k, {a:b for a,b in d.get(k,{}).items() if a in ('ms_per_step','value','error')})
PY
tail -3 $O/bench_cfg2.err
