cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
for v in syn128 syn64; do
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so timeout 300 python -m pytest tests/test_hip_synthesis.py tests/test_hip_requiem.py -m gpu -q -x 2>&1 | tail -1
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in list(k.items())[:3]})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
