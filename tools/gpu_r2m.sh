cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_harvest.py tests/test_hip_longform.py tests/test_hip_edge_cases.py tests/test_hip_requiem.py tests/test_hip_getters.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --config 3 --steps 5 --warmup 2"
$B > $O/bench_default.json 2> $O/bench_default.err
WH_LIB=python-world_amd/lib/variants/libworld_hip_ols_single.so $B > $O/bench_ols_single.json 2> $O/bench_ols_single.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2m/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], {a:b for a,b in list(k.items())[:4]})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
