#!/bin/bash
# round 6, closing: what the two-ballot form of the unit-level early-out costs the scenes the rule keeps it from
# (suzanne: statistic 0.18; closed soups: 0.05) - forced on against the fused test.  Information for LAB.md; the
# threshold stays.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ai; mkdir -p $out
S2="suzanne,1024,128,512,0"; S1="suzanne,1024,64,256,0"
python scripts/quick_bench.py $S2 $S2,seq_unit_ufirst=1 $S1 $S1,seq_unit_ufirst=1 $S2 $S2,seq_unit_ufirst=1 2>&1 | grep "Msamples\|rror" | tee $out/suzanne_two_ballots_forced.txt
