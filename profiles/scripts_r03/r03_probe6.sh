#!/bin/bash
# Round 3, GPU call 6: the headline kernel with guesses taken from the previous pixel's strata.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_round2.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log )
tail -4 $OUT/pytest.log
A=$OUT/spec_ab.txt
: > $A
echo "== shipped build (guesses from the previous pixel's strata)" >> $A
timeout 300 python scripts/quick_bench.py cornell,256,256,256,0 cornell,512,512,256,0 single-sphere,256,256,256,0 example1,256,256,256,0 multi-sphere,256,256,256,0 >> $A 2>&1
echo "== prof build" >> $A
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so timeout 300 python scripts/quick_bench.py cornell,128,128,256,0 >> $A 2>&1
grep -v amdgpu.ids $A
