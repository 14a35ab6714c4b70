cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_cfg2.json 2> gpurun_out/r2a/bench_cfg2.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r2a/bench_cfg2.err
timeout 300 python bench.py --config 3 --steps 5 --warmup 2 > gpurun_out/r2a/bench_cfg3.json 2> gpurun_out/r2a/bench_cfg3.err
timeout 300 python bench.py --config 3 --utts 256 --steps 3 --warmup 1 > gpurun_out/r2a/bench_cfg3_256.json 2> gpurun_out/r2a/bench_cfg3_256.err
timeout 300 python bench.py --config 4 --steps 5 --warmup 2 > gpurun_out/r2a/bench_cfg4.json 2> gpurun_out/r2a/bench_cfg4.err
timeout 400 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r2a/bench_cfg5.json 2> gpurun_out/r2a/bench_cfg5.err
timeout 600 tools/profile_suite.sh 3 r2a/prof_cfg3 > gpurun_out/r2a/prof3.log 2>&1
timeout 600 tools/profile_suite.sh 4 r2a/prof_cfg4 > gpurun_out/r2a/prof4.log 2>&1
ls gpurun_out/r2a
