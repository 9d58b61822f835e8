#!/bin/bash
# round 6 (third session): the unit-level u-first early-out as its own instantiations (traceSequential<...,unit>, picked by
# the dispatcher for scenes whose statistic says so): the tests that name these kernels, ce / suzanne timings with the
# form at its default, forced off and forced on.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ac; mkdir -p $out
( timeout 1500 python -m pytest tests -q -m gpu -k "unit or ties or baseline_scenes or cfg4_full or cfg3_whole or dropped or obj_scene or natural_dispatch or pick" --durations=5 > $out/pytest_subset.log 2>&1; echo "rc=$?" >> $out/pytest_subset.log )
grep -E "passed|failed|rc=" $out/pytest_subset.log | tail -3
for rep in 1 2; do
for sc in ce,2048,8,1024,0 ce,2048,4,256,0 suzanne,1024,128,512,0 suzanne,1024,64,256,0; do
  python scripts/quick_bench.py $sc $sc,seq_unit_ufirst=0 $sc,seq_unit_ufirst=1 2>&1 | grep "Msamples\|rror"
done
done | tee $out/unit_instantiations_ab.txt
