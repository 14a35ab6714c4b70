#!/bin/bash
# round-5 GPU call 8: compact overlap-add rows + wh_ctx_trim: whole suite, the default bench line (memory of the north-star
# block), then the sanitizer harness per path.
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=3 ) > $O/pytest8.log 2>&1
echo "pytest rc=$?" >> $O/pytest8.log
tail -8 $O/pytest8.log | cut -c1-250
( time timeout 1500 python bench.py ) > $O/bench_default_v3.json 2> $O/bench_default_v3.err
echo "bench rc=$?"; tail -c 800 $O/bench_default_v3.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05/bench_default_v3.json") if l.startswith("{")][-1])
    print("ms_per_step", d["ms_per_step"], "one in flight", d["ms_per_step_one_in_flight"], "value", d["value"])
    print("kernel_ms", dict(list(d["kernel_ms"].items())[:8]))
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_ms")})
    print("facade", {k: d.get("facade_batch", {}).get(k) for k in ("resynthesis_flow_ms", "roundtrip_unmodified_ms", "error")})
    ns = d.get("north_star", {})
    print("north_star", {k: ns.get(k) for k in ("ms_per_step", "ms_per_step_one_in_flight", "steps_in_flight", "graph", "x_realtime", "error")})
    print("ns roofline", {k: (ns.get("roofline") or {}).get(k) for k in ("kernel", "frac", "traffic", "traffic_over_algorithmic", "traffic_source")})
    print("ns kernels", ns.get("kernel_ms"))
    print("other", {k: (v.get("ms_per_step"), v.get("ms_per_step_one_in_flight"), v.get("utterances"), v.get("error")) for k, v in d.get("other_configs", {}).items() if isinstance(v, dict)})
    print("survey_8d", d.get("value_survey_8d"), "rt_out_only", d.get("value_roundtrip_out_only"), "cpu", d.get("cpu_baseline", {}).get("value"))
    for k in ("with_transfers", "with_transfers_pipelined", "roundtrip_out_only", "varying_lengths", "decode_alone", "config1_latency", "feature_heads", "swipe"):
        v = d.get(k, {})
        print(k, v.get("error") or v.get("ms_per_step") or {kk: v[kk] for kk in list(v)[:3]})
except Exception as e:
    print("parse failed", e)
PY
bash tools/asan_probe.sh
