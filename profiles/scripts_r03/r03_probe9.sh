#!/bin/bash
# lock-step PERPIXEL kernel: waves per SIMD A/B at the BASELINE shape
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03i
mkdir -p $OUT
cd $REPO
A=$OUT/pix_waves_ab.txt
: > $A
for lib in libptw_hip_pw4.so libptw_hip_pw5.so libptw_hip_pw6.so; do
  echo "== $lib (lock-step kernel)" >> $A
  PTW_LIB_PATH=$REPO/pt-three-ways_amd/$lib PTW_PIX_KERNEL=legacy timeout 300 python scripts/quick_bench.py cornell,1024,1024,256,1 cornell,1024,1024,256,1 suzanne,512,512,64,1 >> $A 2>&1
done
grep -v amdgpu.ids $A
