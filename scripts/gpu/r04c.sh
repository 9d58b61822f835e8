#!/bin/bash
# round 4, third GPU call: where does the decoupled protocol lose?  s_memtime anatomy (prof builds) of
# the decoupled / lock-step / wakeup forms, then timing of the polling variants on sub-runs.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04c; mkdir -p $O
L=$PWD/pt-three-ways_amd
prof() { # name lib scene w h spp rows
  echo "== $1 ($2) $3 $4x$5 @ $6 rows 0:$7" | tee -a $O/anatomy.txt
  PTW_LIB_PATH=$L/$2 timeout 300 python bench.py --scene $3 --width $4 --height $5 --spp $6 --rows 0:$7 --no-cpu-baseline --no-parity --no-secondary 2>&1 \
    | grep -v amdgpu.ids | grep "PHASES\|MASTER\|WORKER" | tee -a $O/anatomy.txt
}
prof decoupled libptw_hip_pwPD.so suzanne 1024 1024 512 8
prof lockstep libptw_hip_pwPL.so suzanne 1024 1024 512 8
prof wakeup3 libptw_hip_pwPW.so suzanne 1024 1024 512 8
prof decoupled libptw_hip_pwPD.so ce 2048 2048 1024 1
prof lockstep libptw_hip_pwPL.so ce 2048 2048 1024 1
run() { # name lib scene w h spp rows
  PTW_LIB_PATH=$L/$2 timeout 600 python bench.py --scene $3 --width $4 --height $5 --spp $6 --rows 0:$7 --no-cpu-baseline --no-parity --no-secondary > $O/$1.json 2> $O/$1.err
  python - "$O/$1.json" "$1" <<'PY' 2>&1 | tee -a gpurun_out/r04c/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f" % r["value"], r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"], "ms/launch %.1f" % r["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for v in "sleep1:libptw_hip.so" "lockstep:libptw_hip_alt.so" "sleep0:libptw_hip_pwS0.so" "wake3:libptw_hip_pwW3.so" "wake8:libptw_hip_pwW8.so"; do
  run suz_${v%%:*} ${v##*:} suzanne 1024 1024 512 256
done
for v in "sleep1:libptw_hip.so" "lockstep:libptw_hip_alt.so" "wake3:libptw_hip_pwW3.so"; do
  run ce_${v%%:*} ${v##*:} ce 2048 2048 1024 8
done
