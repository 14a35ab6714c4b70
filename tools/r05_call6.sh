#!/bin/bash
# round-5 GPU call 6: shipped library (contraction with honoured pragmas): whole suite + short bench; the sanitizer run.
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=3 ) > $O/pytest6.log 2>&1
echo "pytest rc=$?" >> $O/pytest6.log
tail -8 $O/pytest6.log | cut -c1-250
python bench.py --no-extras --no-cpu-baseline --no-pmc > $O/c2e_base.json 2> $O/c2e_base.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/c2e_base.json") if l.startswith("{")][-1])
print("cfg2 %.3f ms/step (one in flight %.3f)" % (d["ms_per_step"], d["ms_per_step_one_in_flight"]), dict(list(d["kernel_ms"].items())[:5]))
PY
python tools/asan/notorch_harness.py 2>&1 | tail -3
bash tools/asan_probe.sh
