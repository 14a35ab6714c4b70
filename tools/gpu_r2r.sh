cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
for v in ab1 ab2 ab3 ab4; do
  WH_LIB=python-world_amd/lib/variants/libworld_hip_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_base.json 2> $O/bench_base.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2r/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {a:round(b,3) for a,b in k.items() if a in ('d4c_kernel','phase_kernel')})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
