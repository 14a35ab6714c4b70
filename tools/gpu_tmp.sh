cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_harvest.py tests/test_hip_longform.py tests/test_hip_edge_cases.py tests/test_hip_fullsize.py tests/test_hip_requiem.py tests/test_hip_getters.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/tmp/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in list(k.items())[:12]})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
