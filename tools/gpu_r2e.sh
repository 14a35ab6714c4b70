cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_d4c.py tests/test_hip_requiem.py tests/test_hip_batch.py tests/test_hip_edge_cases.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_all_fma.so timeout 900 python -m pytest tests -m gpu -q > $O/pytest_fma.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fma.log
tail -8 $O/pytest_fma.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
$B > $O/bench_default.json 2> $O/bench_default.err
for v in d4c_r8c d4c_fma d4c_fma_r8c d4c_wu2 all_fma; do
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so $B > $O/bench_$v.json 2> $O/bench_$v.err
done
WH_LIB=python-world_amd/lib/variants/libworld_hip_d4c_timer.so python tools/d4c_stage_timer.py > $O/stage_timer.txt 2>&1
cat $O/stage_timer.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], 'd4c %.3f'%k.get('d4c_kernel',0), 'resp %.3f'%k.get('response_kernel',0), 'ct %.3f'%k.get('cheaptrick_kernel',0))
    except Exception as e:
        print(f, 'ERR', e)
PY
