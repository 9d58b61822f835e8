#!/bin/bash
# round 6 (third session): the one-master shares by place as shipped - the GPU tests that name the one-master kernels, the
# tie scenes, the OBJ scene; suzanne / ce at 256 passes (default dispatch against the equal shares); the one-master
# prefilter kernels with the same shares (not shipped: measured here).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06y; mkdir -p $out
( timeout 1200 python -m pytest tests -q -m gpu -k "one_master or ties or obj_scene or dispatch or parked or cli or unit or calibrat or streamed" --durations=5 > $out/pytest_subset.log 2>&1; echo "rc=$?" >> $out/pytest_subset.log )
grep -E "passed|failed|rc=" $out/pytest_subset.log | tail -3
S="suzanne,1024,64,256,0"
C="ce,2048,4,256,0"
python scripts/quick_bench.py $S $S,seq_units=3:3:3 $C $C,seq_units=8:8:6 $C,accel=2 $C,accel=2,seq_units=9:6:9 $C,accel=2,seq_units=10:7:7 $S,accel=2 $S,accel=2,seq_units=3:2:1 2>&1 | grep "Msamples\|rror" | tee $out/one_master_shipped.txt
