python -m pytest tests/test_hip_harvest.py tests/test_hip_harvest_rounds.py tests/test_hip_longform.py tests/test_hip_determinism.py tests/test_hip_edge_cases.py -m gpu -x -q > gpurun_out/r06_ab2_tests.log 2>&1; tail -3 gpurun_out/r06_ab2_tests.log
bash tools/bench_variants.sh gpurun_out/ab2_1024 --config 3 --utts 1024 --steps 4 --warmup 1 --in-flight 1 -- base
bash tools/bench_variants.sh gpurun_out/ab2_256 --config 3 --utts 256 --steps 6 --warmup 1 --in-flight 1 -- base
bash tools/profile_suite.sh 3 r06a_cfg3_1024 --utts 1024 --in-flight 1 > /dev/null 2>&1
for d in fetch write; do ls gpurun_out/r06a_cfg3_1024/$d/*/ 2>/dev/null | head -3; done
F=$(ls gpurun_out/r06a_cfg3_1024/fetch/*/*counter_collection.csv | head -1); W=$(ls gpurun_out/r06a_cfg3_1024/write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic_summary.py $F $W 2049024 "config 3 at 1024 x 10 s (Harvest only)" > gpurun_out/r06a_cfg3_1024_hbm_traffic_pmc.txt; head -12 gpurun_out/r06a_cfg3_1024_hbm_traffic_pmc.txt
rm -rf gpurun_out/r06a_cfg3_1024/fetch gpurun_out/r06a_cfg3_1024/write gpurun_out/r06a_cfg3_1024/sqa gpurun_out/r06a_cfg3_1024/sqb
