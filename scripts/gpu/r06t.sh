#!/bin/bash
# round 6 (second session): alignment of branch-only targets, second look: the speculative kernel at 8 / 16 / 128 bytes;
# the single-wave kernel at 16 / 32 / 64 bytes over 256 ... 4096 passes; the one-master worker kernel at 64 bytes.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06t; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2; do
  for v in tree AL3 AL4 AL7; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 200 python scripts/quick_bench.py cornell,512,512,256,0 2>&1 | grep Msamples
  done
  for v in tree G4 G5 G6; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 200 python scripts/quick_bench.py cornell,512,64,256,0,seq_small_kernel=1 cornell,2048,16,512,0,seq_small_kernel=1 cornell,4096,16,1024,0,seq_small_kernel=1 cornell,4096,16,2048,0,seq_small_kernel=1 cornell,4096,16,4096,0,seq_small_kernel=1 2>&1 | grep Msamples
  done
  for v in tree X6; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 200 python scripts/quick_bench.py suzanne,512,128,256,0 ce,512,16,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
