#!/bin/bash
# round 6 (third session): with the unit-level early-out a unit costs less - do the shares by place still sit at
# their optimum?  ce two masters (1024 passes) and one master (256 passes), suzanne with the early-out forced on.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ad; mkdir -p $out
C2="ce,2048,8,1024,0"
C1="ce,2048,4,256,0"
python scripts/quick_bench.py $C2 $C2,seq_units=10:8:9 $C2,seq_units=10:9:8 $C2,seq_units=9:9:9 $C2,seq_units=9:8:10 $C2,seq_units=10:10:7 $C2,seq_units=9:10:8 $C2,seq_units=8:9:10 \
  $C1 $C1,seq_units=9:7:6 $C1,seq_units=8:8:6 $C1,seq_units=9:8:3 $C1,seq_units=10:7:3 $C1,seq_units=8:7:10 $C1,seq_units=10:6:6 2>&1 | grep "Msamples\|rror" | tee $out/shares_with_unit_early_out.txt
