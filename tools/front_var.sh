# tools/front_var.sh <outdir> <variant names...>: config 3 at 64 and 1024 utterances for library variants ("base" = shipped)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  for n in 64 1024; do
    python bench.py --config 3 --utts $n --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/b3_${v}_n$n.json 2> $O/b3_${v}_n$n.err
    python - "$O/b3_${v}_n$n.json" "$v n=$n" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-18s %8.3f ms/step  %s" % (sys.argv[2], d["ms_per_step"], {k: v for k, v in list(d["kernel_ms"].items())[:5]}))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done
