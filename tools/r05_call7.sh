#!/bin/bash
# round-5 GPU call 7: the sanitizer harness, then a full default bench line and the rocprofv3 passes of configs 2 and 4.
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/asan_probe.sh
( time timeout 1200 python bench.py ) > $O/bench_default_v2.json 2> $O/bench_default_v2.err
echo "bench rc=$?"; tail -c 600 $O/bench_default_v2.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05/bench_default_v2.json") if l.startswith("{")][-1])
    print("ms_per_step", d["ms_per_step"], "one in flight", d["ms_per_step_one_in_flight"], "value", d["value"])
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_ms")})
    print("facade", {k: d.get("facade_batch", {}).get(k) for k in ("resynthesis_flow_ms", "roundtrip_unmodified_ms", "error")})
    ns = d.get("north_star", {})
    print("north_star", {k: ns.get(k) for k in ("ms_per_step", "ms_per_step_one_in_flight", "graph", "x_realtime", "error")})
    print("ns roofline", {k: (ns.get("roofline") or {}).get(k) for k in ("kernel", "frac", "traffic", "traffic_over_algorithmic", "traffic_source")})
    print("other", {k: (v.get("ms_per_step"), v.get("ms_per_step_one_in_flight"), v.get("utterances")) for k, v in d.get("other_configs", {}).items() if isinstance(v, dict)})
    print("survey_8d", d.get("value_survey_8d"), "rt_out_only", d.get("value_roundtrip_out_only"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("parse failed", e)
PY
for c in 2 4; do timeout 900 tools/profile_suite.sh $c r05/prof_cfg$c --in-flight 1 > $O/prof$c.log 2>&1; echo "profile cfg$c rc=$?"; done
