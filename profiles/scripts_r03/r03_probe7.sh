#!/bin/bash
# Round 3, GPU call: where does a tick of the two-master protocol go?  (prof build, suzanne / ce)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
cd $REPO
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so timeout 600 python scripts/quick_bench.py suzanne,64,64,512,0 suzanne,64,64,256,0 ce,32,32,1024,0 > $OUT/master_phases.txt 2>&1
grep -v amdgpu.ids $OUT/master_phases.txt
