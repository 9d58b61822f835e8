#!/bin/bash
# round 6 (third session): the unit-level u-first early-out in the worker waves (testTriangleUnit) against the fused test
# (variant build -DPTW_SEQ_UNIT_UFIRST=0), same box, alternating; then the parity tests that hold these kernels to the oracle.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06aa; mkdir -p $out
A="ce,2048,8,1024,0 ce,2048,4,256,0 suzanne,1024,128,512,0 suzanne,1024,64,256,0"
for rep in 1 2; do
  for lib in libptw_hip.so libptw_hip_pwnou.so; do
    echo "== $lib"
    PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib python scripts/quick_bench.py $A 2>&1 | grep "Msamples\|rror"
  done
done | tee $out/unit_ufirst_ab.txt
for lib in libptw_hip.so libptw_hip_pwnou.so; do
  echo "== $lib"
  PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib SWEEP_SIZES=1000,1900,3400 SWEEP_PASSES=256,512 SWEEP_POLICIES=0 timeout 600 python scripts/dispatch_sweep.py $out/sweep_$lib.md > $out/sweep_$lib.log 2>&1
  grep "sequential" $out/sweep_$lib.md | grep "one master\|two masters" | cut -d'|' -f3,4,6,7
done | tee $out/unit_ufirst_soups.txt
( timeout 1500 python -m pytest tests -q -m gpu -k "cfg3_whole or cfg4_full or two_master or one_master or ties or obj_scene or streamed or parity" --durations=5 > $out/pytest_subset.log 2>&1; echo "rc=$?" >> $out/pytest_subset.log )
grep -E "passed|failed|rc=" $out/pytest_subset.log | tail -3
