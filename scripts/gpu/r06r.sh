#!/bin/bash
# round 6 (second session): does the LAYOUT of the code decide?  Variant K's source and S1's (ten spill operations
# fewer per round, yet 3 % slower) with branch-only targets aligned to 32 / 64 bytes (-mllvm
# -align-all-nofallthru-blocks=5 / 6), loops aligned to 64 bytes (-falign-loops=64), all blocks to 8 bytes.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06r; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in K AL5 AL6 LP6 AB3 S1 S1AL5 S1AL6; do
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
