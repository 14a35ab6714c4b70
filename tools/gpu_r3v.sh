cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_harvest.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
