#!/bin/bash
# round 6 (second session): suzanne's 16 units of 64 triangles over the six worker waves of the two-master kernel -
# the shares by place (older wave of a pair : younger : master-side) the dispatcher gives (3:3:3) against lighter
# master-side waves (the tick is the masters': their SIMD neighbours' load is what they compete with).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06v; mkdir -p $out
for rep in 1 2; do
  python scripts/quick_bench.py suzanne,1024,128,512,0 suzanne,1024,128,512,0,seq_units=3:3:2 suzanne,1024,128,512,0,seq_units=4:2:2 suzanne,1024,128,512,0,seq_units=4:3:1 suzanne,1024,128,512,0,seq_units=3:2:3 suzanne,1024,128,512,0,seq_units=4:4:0 suzanne,1024,128,512,0,seq_units=5:3:0 2>&1 | grep Msamples
done > $out/ab.txt 2>&1
cat $out/ab.txt
