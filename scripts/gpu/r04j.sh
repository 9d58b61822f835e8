#!/bin/bash
# round 4: traceSequentialSpec with the many-candidates reduction as one LDS atomic (-DPTW_SPEC_LDS_MIN=1,
# libptw_hip_pwSL.so) against the shipped DPP form: parity, then A/B on the headline scene (512 x 512 @ 256).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04j; mkdir -p $O
L=$PWD/pt-three-ways_amd
PTW_LIB_PATH=$L/libptw_hip_pwSL.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu > $O/pytest_sl.log 2>&1; echo "pytest_sl rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_sl.log | tee -a $O/summary.txt
run() { local name=$1 lib=$2; shift 2
  PTW_LIB_PATH=$L/$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-parity --no-secondary --no-other-configs --no-strict > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY' 2>&1 | tee -a gpurun_out/r04j/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f" % r["value"], r["roofline"]["kernel"], "ms/launch %.1f" % r["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2 3; do
  run dpp_$i libptw_hip.so --width 512 --height 512
  run lds_$i libptw_hip_pwSL.so --width 512 --height 512
done
