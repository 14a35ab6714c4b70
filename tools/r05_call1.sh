#!/bin/bash
# round-5 GPU call 1: the whole -m gpu suite (new: quoted sizes, lazy dicts, 2-rank rehearsal, flag/time-base contracts) and
# the default bench line.  usage (through gpurun): bash tools/r05_call1.sh
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest1.log 2>&1
echo "pytest rc=$?" >> $O/pytest1.log
tail -5 $O/pytest1.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05/bench_default.json") if l.startswith("{")][-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    print("kernel_ms", dict(list(d["kernel_ms"].items())[:6]))
    print("facade", d.get("facade_batch"))
    ns = d.get("north_star", {})
    print("north_star", {k: ns.get(k) for k in ("ms_per_step", "graph", "eager_ms_per_step", "x_realtime", "error")})
    print("ns roofline", ns.get("roofline"))
    print("other", {k: (v.get("ms_per_step"), v.get("utterances")) for k, v in d.get("other_configs", {}).items() if isinstance(v, dict)})
    print("survey_8d", d.get("value_survey_8d"), "rt_out_only", d.get("value_roundtrip_out_only"))
except Exception as e:
    print("parse failed", e)
PY
