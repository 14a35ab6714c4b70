cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
for v in ols2k ols2kw4 ols4kw4; do
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so timeout 300 python -m pytest tests/test_hip_harvest.py -m gpu -q -x 2>&1 | tail -1
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in list(k.items())[:3]}, {a:round(b,3) for a,b in k.items() if 'fft' in a})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
