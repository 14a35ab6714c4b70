cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --config 3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2h/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], {a:b for a,b in list(k.items())[:7]})
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 600 tools/profile_suite.sh 3 r2h/prof_cfg3 > $O/prof3.log 2>&1
