#!/bin/bash
# round-5 GPU call 4: FMA contraction in the spectral kernels (D4C, CheapTrick, pulse responses: build variants), parity of
# the whole suite on the combined variant, benches per variant; then the sanitizer pass/abort run.
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
FL="--no-extras --no-cpu-baseline --no-pmc"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms"]
    pick = {n: k[n] for n in ("d4c_kernel", "response_kernel", "cheaptrick_kernel", "req_filter_kernel", "band_events_kernel") if n in k}
    print("%-22s %8.3f ms/step (one in flight %.3f)  %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_one_in_flight") or 0, pick))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
( time WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_fma3.so timeout 1500 python -m pytest tests -m gpu -q --durations=3 ) > $O/pytest4_fma3.log 2>&1
echo "pytest rc=$?" >> $O/pytest4_fma3.log
tail -30 $O/pytest4_fma3.log | cut -c1-250
for rep in 1 2; do
for v in base d4cfma ctfma synfma fma3; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py $FL > $O/c2c_${v}_$rep.json 2> $O/c2c_${v}_$rep.err; show $O/c2c_${v}_$rep.json cfg2_${v}_$rep
done
done
for v in base fma3; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py $FL --config 4 --steps 10 > $O/c4c_$v.json 2> $O/c4c_$v.err; show $O/c4c_$v.json cfg4_$v
  python bench.py $FL --config 5 --steps 3 > $O/c5c_$v.json 2> $O/c5c_$v.err; show $O/c5c_$v.json cfg5_$v
done
unset WH_LIB
bash tools/asan_probe.sh
