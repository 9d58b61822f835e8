#!/bin/bash
# round 6 (third session): shares by the wave's place for the ONE-master kernels (at most one pass per CU: suzanne / ce at
# 256 passes - and every per-GPU share of cfg3 / cfg4 on eight GPUs): worker waves 1-3 are the older waves of SIMD 1-3,
# 5-7 the younger ones, wave 4 sits beside the master.  seq_units = older : younger : master-side.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06y; mkdir -p $out
S="suzanne,1024,64,256,0"
C="ce,2048,4,256,0"
for rep in 1 2; do
python scripts/quick_bench.py $S $S,seq_units=3:2:1 $S,seq_units=3:2:2 $S,seq_units=3:3:0 $S,seq_units=4:2:0 $S,seq_units=4:1:1 $S,seq_units=3:2:3 $S,seq_units=2:2:4 $S,seq_units=3:1:4 \
  $C $C,seq_units=8:8:6 $C,seq_units=8:7:9 $C,seq_units=9:7:6 $C,seq_units=10:6:6 $C,seq_units=9:6:9 $C,seq_units=8:6:12 $C,seq_units=10:8:0 2>&1 | grep "Msamples\|rror"
done | tee $out/one_master_shares.txt
