#!/bin/bash
# HBM traffic per kernel of one bench config (GPU box, through gpurun): the two PMC passes the guide prescribes
# (FETCH_SIZE and WRITE_SIZE separately) and their digest.
#   tools/traffic_only.sh <config> <tag> <frames per launch> <label> [digest key=value ...] -- [bench flags]
set -u
CFG=$1; TAG=$2; FR=$3; LABEL=$4; shift 4
KV=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do KV+=("$1"); shift; done
[ $# -gt 0 ] && shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-graph $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- $B > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- $B > /dev/null 2> $OUT/write.err
python tools/pmc_traffic_summary.py $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv $FR "$LABEL" "${KV[@]}" > gpurun_out/${TAG}_hbm_traffic_pmc.txt
rm -rf $OUT/fetch $OUT/write
head -14 gpurun_out/${TAG}_hbm_traffic_pmc.txt
