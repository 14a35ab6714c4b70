cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/tmp; mkdir -p $O; rm -f $O/bench_*.json
for v in love8; do
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so timeout 200 python -m pytest tests/test_hip_d4c.py tests/test_hip_requiem.py -m gpu -q -x 2>&1 | tail -1
WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/tmp/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in k.items() if a in ('love_train_kernel','d4c_kernel','hv_prune_kernel','hc_base_kernel')})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
