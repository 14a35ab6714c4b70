#!/bin/bash
# round-5 GPU call 2: the overlap-add without atomics (rows + gather) — the whole -m gpu suite on it — then short benches:
# config 2 with two steps in flight (and one), config 4, and the build variants (D4C centroid loop, Requiem run lengths).
O=gpurun_out/r05
mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > $O/pytest2.log 2>&1
echo "pytest rc=$?" >> $O/pytest2.log
tail -25 $O/pytest2.log
FL="--no-extras --no-cpu-baseline --no-pmc"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-22s %8.3f ms/step (one in flight %s)  %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_one_in_flight"), {k: v for k, v in list(d["kernel_ms"].items())[:6]}))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
for rep in 1 2; do
  python bench.py $FL > $O/c2_base_$rep.json 2> $O/c2_base_$rep.err; show $O/c2_base_$rep.json base_cfg2_$rep
  WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_centloop.so python bench.py $FL > $O/c2_centloop_$rep.json 2> $O/c2_centloop_$rep.err; show $O/c2_centloop_$rep.json centloop_cfg2_$rep
done
python bench.py $FL --in-flight 1 > $O/c2_base_if1.json 2> $O/c2_base_if1.err; show $O/c2_base_if1.json base_cfg2_inflight1
python bench.py $FL --in-flight 3 > $O/c2_base_if3.json 2> $O/c2_base_if3.err; show $O/c2_base_if3.json base_cfg2_inflight3
for v in base runf4 runf1 runf16; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py $FL --config 4 --steps 10 > $O/c4_$v.json 2> $O/c4_$v.err; show $O/c4_$v.json cfg4_$v
done
unset WH_LIB
python bench.py $FL --config 3 --steps 10 > $O/c3_base.json 2> $O/c3_base.err; show $O/c3_base.json cfg3_base
python bench.py $FL --config 5 --steps 3 > $O/c5_base.json 2> $O/c5_base.err; show $O/c5_base.json cfg5_base
WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_centloop.so python bench.py $FL --config 5 --steps 3 > $O/c5_centloop.json 2> $O/c5_centloop.err; show $O/c5_centloop.json cfg5_centloop
