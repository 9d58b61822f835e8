#!/bin/bash
# round 6 (third session): the one-master kernels with the shares by place as the rule (seqUnitSplitByPlaceOneMaster) and the
# new <9,7> / <10,7> instantiations: suzanne and ce at 256 passes, the forced neighbours, closed soups in between.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06y; mkdir -p $out
S="suzanne,1024,64,256,0"
C="ce,2048,4,256,0"
for rep in 1 2; do
python scripts/quick_bench.py $S $S,seq_units=3:3:3 $S,seq_units=3:2:1 \
  $C $C,seq_units=8:8:6 $C,seq_units=9:6:9 $C,seq_units=10:7:3 $C,seq_units=9:7:6 $C,seq_units=10:6:6 $C,seq_units=9:6:10 $C,seq_units=8:7:10 $C,seq_units=9:5:10 $C,seq_units=10:5:9 2>&1 | grep "Msamples\|rror"
done | tee $out/one_master_shares_rule.txt
SWEEP_SIZES=512,700,800,1000,1100,1300,1600,1900,2400,3000,3600 SWEEP_PASSES=256 SWEEP_POLICIES=0 timeout 900 python scripts/dispatch_sweep.py $out/sweep_one_master_shares_by_place.md > $out/sweep.log 2>&1
grep "sequential" $out/sweep_one_master_shares_by_place.md | grep -v "two masters" | cut -d'|' -f3,4,6,7,9,10,11,12
