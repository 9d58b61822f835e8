#!/bin/bash
# round 6 (second session): the tree with the generator wave adding the committed radiance (the form variant K
# measured, hooks removed): the whole GPU suite, the instrumented build's anatomy, and K against the tree's library.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06n; mkdir -p $out
L=$PWD/pt-three-ways_amd
( timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log )
grep -E "passed|failed|rc=" $out/pytest_gpu.log | tail -3
for rep in 1 2; do
  for v in tree K; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 300 python scripts/quick_bench.py cornell,64,64,256,0 > $out/spec_histogram.txt 2>&1
grep "SPEC wave" $out/spec_histogram.txt | head -12
