# tools/var_cfg.sh <outdir> <config> <variants...>: one bench config (10 steps) per library variant ("base" = shipped), one line each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; C=$1; shift; mkdir -p $O
for v in "$@"; do
  if [ "$v" = "base" ]; then unset WH_LIB; else export WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_$v.so; fi
  python bench.py --config $C --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-pmc > $O/b${C}_$v.json 2> $O/b${C}_$v.err
  python - "$O/b${C}_$v.json" "$v cfg$C" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %8.3f ms/step  %s" % (sys.argv[2], d["ms_per_step"], {k: v for k, v in list(d["kernel_ms"].items())[:5]}))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
done
