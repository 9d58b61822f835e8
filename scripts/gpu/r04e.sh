#!/bin/bash
# round 4, fifth GPU call: the master's pick as "minimum, then lowest index among equals" (PICK_MIN) and
# the worker waves' many-candidates reduction as one LDS atomic (LDS_MIN): the whole suite, then A/B on
# one box against both off (alt), each alone (pwPM: PICK_MIN only, pwLM: LDS_MIN only).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04e; mkdir -p $O
L=$PWD/pt-three-ways_amd
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest_gpu.log | grep "passed\|failed\|Error" | tee -a $O/summary.txt
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  PTW_LIB_PATH=$L/$lib timeout 600 python bench.py "$@" --no-cpu-baseline --no-parity --no-secondary > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY' 2>&1 | tee -a gpurun_out/r04e/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f" % r["value"], r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"], "ms/launch %.1f" % r["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
SUZ="--scene suzanne --width 1024 --height 1024 --spp 512 --rows 0:192"
CE="--scene ce --width 2048 --height 2048 --spp 1024 --rows 0:8"
SUZ1="--scene suzanne --width 1024 --height 1024 --spp 256 --rows 0:192"
for v in "both:libptw_hip.so" "neither:libptw_hip_alt.so" "pick_min:libptw_hip_pwPM.so" "lds_min:libptw_hip_pwLM.so" "both_2:libptw_hip.so"; do
  run suz_${v%%:*} ${v##*:} $SUZ
done
for v in "both:libptw_hip.so" "neither:libptw_hip_alt.so" "pick_min:libptw_hip_pwPM.so" "lds_min:libptw_hip_pwLM.so"; do
  run ce_${v%%:*} ${v##*:} $CE
done
for v in "both:libptw_hip.so" "neither:libptw_hip_alt.so"; do
  run suz256_${v%%:*} ${v##*:} $SUZ1
done
