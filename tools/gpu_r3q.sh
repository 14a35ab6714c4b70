cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_harvest.py tests/test_hip_requiem.py tests/test_hip_edge_cases.py tests/test_hip_getters.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
