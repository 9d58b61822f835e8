#!/bin/bash
# round 4, fourth GPU call: the whole GPU suite on the tree as it stands (lock step shipped, decoupled
# in the experiments build), then same-box A/B of the scalar-constant change (libptw_hip_pwNS.so =
# -DPTW_SCONST=0) on the three kernel families, and the carried-surface lock-step PERPIXEL variant.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04d; mkdir -p $O
L=$PWD/pt-three-ways_amd
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log | tee -a $O/summary.txt
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  PTW_LIB_PATH=$L/$lib timeout 600 python bench.py "$@" --no-cpu-baseline --no-parity > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY' 2>&1 | tee -a gpurun_out/r04d/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f" % r["value"], r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"], "ms/launch %.1f" % r["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
HEAD="--width 512 --height 512 --spp 256 --no-secondary --no-other-configs --no-strict"
for i in 1 2; do
  run head_sconst_$i libptw_hip.so $HEAD
  run head_nosconst_$i libptw_hip_pwNS.so $HEAD
done
SUZ="--scene suzanne --width 1024 --height 1024 --spp 512 --rows 0:192 --no-secondary"
run suz_sconst libptw_hip.so $SUZ
run suz_nosconst libptw_hip_pwNS.so $SUZ
CE="--scene ce --width 2048 --height 2048 --spp 1024 --rows 0:8 --no-secondary"
run ce_sconst libptw_hip.so $CE
run ce_nosconst libptw_hip_pwNS.so $CE
PIX="--policy perpixel --steps 3 --warmup 1"
run pix_sconst libptw_hip.so $PIX
run pix_nosconst libptw_hip_pwNS.so $PIX
run pix_carried libptw_hip_pwPR0.so $PIX
run pix_sconst_2 libptw_hip.so $PIX
run pixsuz_sconst libptw_hip.so --scene suzanne --spp 64 --policy perpixel --steps 2 --warmup 1
run pixsuz_nosconst libptw_hip_pwNS.so --scene suzanne --spp 64 --policy perpixel --steps 2 --warmup 1
