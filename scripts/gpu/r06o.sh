#!/bin/bash
# which of three "dead code" clean-ups of variant K costs 1.6 %?  Vr: K's source; Va: ray counter store merged into the
# park block; Vb: histogram halving inside the refresh block; Vc: the unused `result` sum removed; Vt: all three.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06o; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in Vr Va Vb Vc Vt K; do
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
