cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_longform.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
WH_LIB=python-world_amd/lib/variants/libworld_hip_d4c_timer.so python tools/d4c_stage_timer.py > $O/stage_timer.txt 2>&1
cat $O/stage_timer.txt
timeout 600 tools/profile_suite.sh 2 r2c/prof_cfg2 > $O/prof2.log 2>&1
