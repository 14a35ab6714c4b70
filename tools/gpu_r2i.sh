cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
timeout 600 tools/profile_suite.sh 3 r2i/prof_cfg3 > $O/prof3.log 2>&1
timeout 600 tools/profile_suite.sh 4 r2i/prof_cfg4 > $O/prof4.log 2>&1
python bench.py --config 3 --utts 256 --steps 3 --warmup 1 > $O/bench_cfg3_256.json 2> $O/bench_cfg3_256.err
python bench.py --config 3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
