#!/bin/bash
# round-5 GPU call 3: the reworked gathers / Requiem rows (whole suite), their build variants, the facade block, then the
# sanitizer probe (last: whatever it does to the box cannot cost the measurements before it).
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 ) > $O/pytest3.log 2>&1
echo "pytest rc=$?" >> $O/pytest3.log
tail -12 $O/pytest3.log
FL="--no-extras --no-cpu-baseline --no-pmc"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms"]
    pick = {n: k[n] for n in ("d4c_kernel", "response_kernel", "response_gather_kernel", "req_filter_kernel", "req_gather_kernel", "peak_max_kernel") if n in k}
    print("%-22s %8.3f ms/step (one in flight %.3f)  %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_one_in_flight") or 0, pick))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
for v in base gper2 gper8; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py $FL > $O/c2b_$v.json 2> $O/c2b_$v.err; show $O/c2b_$v.json cfg2_$v
done
for v in base reqw8 runf2 runf8 runf1; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py $FL --config 4 --steps 10 > $O/c4b_$v.json 2> $O/c4b_$v.err; show $O/c4b_$v.json cfg4_$v
done
unset WH_LIB
python - <<'PY' 2>&1 | tail -5
import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "python-world_amd")
import torch, bench
xs = bench.make_inputs(0, 64, 16000, 10.0)
print("facade_batch", json.dumps(bench.facade_batch_block(torch, xs, 16000)))
PY
bash tools/asan_probe.sh
