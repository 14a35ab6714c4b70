# Full GPU pass (through gpurun from the repo root): every -m gpu test, every bench config, and the rocprofv3 passes of
# configs 2-4 (tools/profile_suite.sh); results under gpurun_out/full/.  Digest with tools/pmc_traffic_summary.py and
# tools/sq_counters_summary.py into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/full; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config 3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 3 --utts 256 --steps 3 --warmup 1 > $O/bench_cfg3_256.json 2> $O/bench_cfg3_256.err
python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config 5 --steps 2 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --lanes 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_cfg2_lanes2.json 2> $O/bench_cfg2_lanes2.err
for c in 2 3 4; do timeout 600 tools/profile_suite.sh $c full/prof_cfg$c > $O/prof$c.log 2>&1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/full/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['kernel_ms']
        print(f.split('/')[-1], 'ms/step %.2f'%d['ms_per_step'], 'xRT %.0f'%d['x_realtime'], 'graph', d['graph'], {a:b for a,b in list(k.items())[:4]})
        for key in ('with_transfers','with_transfers_overlapped','north_star'):
            if key in d: print('   ', key, {kk: d[key].get(kk) for kk in ('ms_per_step','x_realtime','error')})
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
