# Re-take the config 3 / 4 profile passes at the final state of the round (after the Harvest kernel changes).
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_edge_cases.py -m gpu -x -q 2>&1 | tail -2
for c in 3 4; do timeout 900 tools/profile_suite.sh $c r4p2/cfg$c > gpurun_out/r4p2_cfg$c.log 2>&1; done
du -sh gpurun_out/r4p2
