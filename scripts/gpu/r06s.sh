#!/bin/bash
# round 6 (second session): branch-only targets aligned to 32 / 64 bytes in the two-master kernels (cfg3, cfg4: eight
# waves per CU) and in the single-wave kernel (cfg5's form: one wave per SIMD at 1024 passes) - does layout matter there?
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06s; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2; do
  for v in tree W5 W6; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 200 python scripts/quick_bench.py suzanne,1024,128,512,0 ce,2048,8,1024,0 2>&1 | grep Msamples
  done
  for v in tree G6; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 200 python scripts/quick_bench.py cornell,4096,16,1024,0,seq_small_kernel=1 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
