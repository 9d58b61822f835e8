#!/bin/bash
# round 5: the whole GPU suite on the tree with the kernel-argument diet (camera in LDS, answers read as
# b128), and a same-box A/B against the library of commit 29a2222 (before it)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05i}; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log ); grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
for rep in 1 2; do
for lib in libptw_hip.so libptw_hip_pwprev.so; do
  echo "== $lib (run $rep)"
  PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib timeout 300 python scripts/quick_bench.py \
    cornell,1024,64,256,0 suzanne,256,256,512,0 suzanne,256,256,256,0 ce,2048,8,1024,0 cornell,512,512,32,1 2>&1 | grep -v amdgpu.ids
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
