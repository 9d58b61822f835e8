#!/bin/bash
# round 6 (second session): the shares by place (the younger wave of a worker pair gets ~70 % of an older wave's units)
# below the 31 units from which the dispatcher applies them: closed soups of 6 ... 30 units, suzanne (16 units).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06w; mkdir -p $out
SWEEP_SIZES=384,512,700,1000,1300,1600,1900 SWEEP_PASSES=512 timeout 900 python scripts/dispatch_sweep.py $out/sweep_shares_by_place_small_unit_counts.md > $out/sweep.log 2>&1
grep "sequential" $out/sweep_shares_by_place_small_unit_counts.md | cut -d'|' -f3,4,6,7,9,10,11,12
python scripts/quick_bench.py suzanne,1024,128,512,0 suzanne,1024,128,512,0,seq_units=3:2:3 suzanne,1024,128,512,0,seq_units=4:1:3 suzanne,1024,128,512,0,seq_units=3:1:4 suzanne,1024,128,512,0,seq_units=2:2:4 2>&1 | grep Msamples > $out/suzanne_shares.txt
cat $out/suzanne_shares.txt
