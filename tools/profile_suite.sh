#!/bin/bash
# rocprofv3 passes for one bench config on the GPU box (run through gpurun from the repo root):
#   tools/profile_suite.sh <config> <tag> [extra bench flags]
# Writes gpurun_out/<tag>/{trace,fetch,write,sqa,sqb}/...csv: kernel-trace stats and four PMC passes (FETCH_SIZE and
# WRITE_SIZE in their own passes as MI355X_MICROARCH.md prescribes; PMC never combined with the trace domains gpurun
# refuses).  Digest them afterwards with tools/pmc_traffic_summary.py and tools/sq_counters_summary.py.
set -u
CFG=$1; TAG=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-graph $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $B > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- $B > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- $B > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sqa -o a --output-format csv -- $B > /dev/null 2> $OUT/sqa.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $OUT/sqb -o b --output-format csv -- $B > /dev/null 2> $OUT/sqb.err
# keep only what the digests need (the merge back is capped at 64 MiB)
find $OUT -name "*agent_info*" -delete 2>/dev/null
ls -R $OUT | head -40
