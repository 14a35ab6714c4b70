cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_dio.py tests/test_hip_harvest.py tests/test_hip_synthesis.py tests/test_hip_cumsum.py tests/test_hip_edge_cases.py tests/test_hip_batch.py tests/test_hip_longform.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for v in x4k256 x4k512 x8k512 x8k1024 x2k512; do
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
done
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2q/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in k.items() if a in ('phase_kernel','iir_fwd_kernel','iir_bwd_kernel','contour_kernel','prep_kernel','band_events_kernel','hv_iir_fwd_kernel','hv_iir_bwd_kernel','lowcut_kernel')})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
