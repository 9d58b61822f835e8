#!/bin/bash
# lock-step PERPIXEL kernel: first-bounce surface rebuilt per sub-sample (10 spilled registers) against carried (80)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03j
mkdir -p $OUT
cd $REPO
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_round2.py tests/test_gpu_accel.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log )
( timeout 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "perpixel" >> $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log )
grep -E "passed|failed|rc=" $OUT/pytest.log
A=$OUT/rebuild_ab.txt
: > $A
for lib in libptw_hip.so libptw_hip_alt.so libptw_hip.so; do
  echo "== $lib (lock-step kernel)" >> $A
  PTW_LIB_PATH=$REPO/pt-three-ways_amd/$lib PTW_PIX_KERNEL=legacy timeout 300 python scripts/quick_bench.py cornell,1024,1024,256,1 cornell,1024,1024,256,1 suzanne,512,512,64,1 single-sphere,512,512,64,1 >> $A 2>&1
done
grep -v amdgpu.ids $A
