cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06s3_cfg5; mkdir -p $O/digests
python bench.py --config 5 --steps 4 --warmup 1 --no-pmc > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 tools/profile_suite.sh 5 r06s3_cfg5/prof_cfg5 --in-flight 1 > $O/prof5.log 2>&1; echo "profile cfg5 rc=$?"
d=$O/prof_cfg5
cp $d/trace/t_kernel_stats.csv $O/digests/cfg5_kernel_stats.csv
python tools/pmc_traffic_summary.py $d/fetch/f_counter_collection.csv $d/write/w_counter_collection.csv 192016 "config 5 (16 x 60 s, 48 kHz)" fs=48000 fft=2048 out_hop_scale=2 > $O/digests/cfg5_hbm_traffic_pmc.txt 2>> $O/digest.err
python tools/sq_counters_summary.py $d/sqa/a_counter_collection.csv $d/sqb/b_counter_collection.csv $d/trace/t_kernel_trace.csv "config 5" > $O/digests/cfg5_sq_counters.txt 2>> $O/digest.err
rm -rf $d/fetch $d/write $d/sqa $d/sqb $d/trace/t_kernel_trace.csv
head -5 $O/digests/cfg5_sq_counters.txt; grep response $O/digests/cfg5_hbm_traffic_pmc.txt | head -3
