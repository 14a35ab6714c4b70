cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
B="python bench.py --config 3 --steps 5 --warmup 2"
$B > $O/bench_default.json 2> $O/bench_default.err
WH_LIB=python-world_amd/lib/variants/libworld_hip_ols_pf2.so $B > $O/bench_ols_pf2.json 2> $O/bench_ols_pf2.err
WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_ols_pf2.so python -m pytest tests/test_hip_harvest.py -m gpu -q 2>&1 | tail -2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2n/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], {a:b for a,b in list(k.items())[:4]})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
