cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
for v in ft128_r8 ft128_r4 ft128_r8c; do
  WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so timeout 300 python -m pytest tests/test_hip_d4c.py tests/test_hip_batch.py -m gpu -x -q 2>&1 | tail -2
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so $B > $O/bench_$v.json 2> $O/bench_$v.err
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --config 5 --steps 2 --warmup 1 --no-graph > $O/bench5_$v.json 2> $O/bench5_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/bench*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], 'd4c %.3f'%k.get('d4c_kernel',0), 'resp %.3f'%k.get('response_kernel',0), 'ct %.3f'%k.get('cheaptrick_kernel',0))
    except Exception as e:
        print(f, 'ERR', e)
PY
