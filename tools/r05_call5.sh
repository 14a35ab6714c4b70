#!/bin/bash
# round-5 GPU call 5: the shipped library with FMA contraction in the spectral kernels (whole suite, bench), and the
# sanitizer run with the reports visible (pytest -s: its fd capture swallowed them in call 4).
O=gpurun_out/r05
mkdir -p $O gpurun_out/asan
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=3 ) > $O/pytest5.log 2>&1
echo "pytest rc=$?" >> $O/pytest5.log
tail -8 $O/pytest5.log | cut -c1-250
FL="--no-extras --no-cpu-baseline --no-pmc"
python bench.py $FL > $O/c2d_base.json 2> $O/c2d_base.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/c2d_base.json") if l.startswith("{")][-1])
print("cfg2 %.3f ms/step (one in flight %.3f)" % (d["ms_per_step"], d["ms_per_step_one_in_flight"]), dict(list(d["kernel_ms"].items())[:5]))
PY
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0:allocator_may_return_null=1:halt_on_error=1
export LD_LIBRARY_PATH=$(dirname "$RT"):$LD_LIBRARY_PATH WH_LIB=$PWD/python-world_amd/lib/variants/libworld_hip_asan.so
A=gpurun_out/asan
for t in "tests/test_hip_dio.py::test_stonemask_vs_golden" "tests/test_hip_cheaptrick.py" "tests/test_hip_d4c.py" "tests/test_hip_synthesis.py" "tests/test_hip_requiem.py" "tests/test_hip_harvest.py" "tests/test_hip_determinism.py" "tests/test_hip_edge_cases.py" "tests/test_hip_swipe.py" "tests/test_hip_features.py" "tests/test_hip_modifiers.py"; do
  n=$(echo $t | tr '/:.' '___')
  timeout 600 python -m pytest "$t" -m gpu -x -s -q > $A/s_$n.log 2>&1
  echo "$t rc=$? : $(grep -a -m1 -E 'ERROR: AddressSanitizer|Hostcall|passed|failed|Fatal' $A/s_$n.log | cut -c1-200)"
done
for f in $A/s_*.log; do if grep -a -q "AddressSanitizer" $f; then echo "=== $f"; grep -a -A25 -m1 "ERROR: AddressSanitizer" $f | cut -c1-220; fi; done 2>/dev/null | head -120
