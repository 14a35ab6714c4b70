# tools/front_env.sh <outdir> <utts> VAR=val[,VAR=val] ...: config 3 with environment settings, one line each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; shift; N=$1; shift; mkdir -p $O
for e in "$@"; do
  tag=$(echo "$e" | tr '=,' '__')
  env $(echo "$e" | tr ',' ' ') python bench.py --config 3 --utts $N --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/b3_${tag}_n$N.json 2> $O/b3_${tag}_n$N.err
  python - "$O/b3_${tag}_n$N.json" "$e n=$N" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms"]
    print("%-34s %8.3f ms/step  front %.2f band_events %.3f raw %.3f detect %.2f" % (sys.argv[2], d["ms_per_step"], k.get("hv_front_kernel", 0), k.get("band_events_kernel", 0), k.get("hv_raw_kernel", 0), k.get("hv_detect_kernel", 0)))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
