# Round-4 profile set (through gpurun): rocprofv3 kernel stats + PMC passes of configs 2-5, and config 3 once more with
# the fused Harvest front end (WH_HV_FRONT=1) for the traffic comparison.  Digest with tools/r04_digest.sh.
cd $GRAFT_REPO_ROOT
for c in 2 3 4 5; do timeout 900 tools/profile_suite.sh $c r4p/cfg$c > gpurun_out/r4p_cfg$c.log 2>&1; done
WH_HV_FRONT=1 timeout 900 tools/profile_suite.sh 3 r4p/cfg3_front > gpurun_out/r4p_cfg3_front.log 2>&1
WH_HV_FRONT=1 timeout 900 tools/profile_suite.sh 4 r4p/cfg4_front > gpurun_out/r4p_cfg4_front.log 2>&1
du -sh gpurun_out/r4p
