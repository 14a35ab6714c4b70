cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_cheaptrick.py tests/test_hip_requiem.py tests/test_hip_batch.py tests/test_hip_edge_cases.py tests/test_hip_features.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3u/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in list(k.items())[:5]})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
