cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
$B > $O/bench_default.json 2> $O/bench_default.err
for v in run4 run8 run32 run64; do
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so $B > $O/bench_$v.json 2> $O/bench_$v.err
done
python bench.py --config 5 --steps 2 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2g/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], 'graph',d['graph'], 'd4c %.3f'%k.get('d4c_kernel',0), 'resp %.3f'%k.get('response_kernel',0), 'reqf %.3f'%k.get('req_filter_kernel',0))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 600 tools/profile_suite.sh 2 r2g/prof_cfg2 > $O/prof2.log 2>&1
