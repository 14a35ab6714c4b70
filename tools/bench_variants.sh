#!/bin/bash
# tools/bench_variants.sh <outdir> <bench flags...> -- <variant names ("" = shipped library)>: one short bench per variant
# on the same box, printing ms/step and the three heaviest kernels.
O=$1; shift
FL=()
while [ "$1" != "--" ]; do FL+=("$1"); shift; done
shift
mkdir -p $O
for v in "$@"; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py "${FL[@]}" --no-extras --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err
  python - "$O/b_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-10s %8.3f ms/step  %s" % (sys.argv[2], d["ms_per_step"], {k: v for k, v in list(d["kernel_ms"].items())[:4]}))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
