#!/bin/bash
# Round 3, GPU call: the masters' inline miss path + 16-byte answers; parity, A/B, phases.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03h
mkdir -p $OUT
cd $REPO
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log )
tail -3 $OUT/pytest.log
W=$OUT/master_ab.txt
: > $W
echo "== shipped build (inline miss path, 16-byte answers)" >> $W
timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $W 2>&1
echo "== alt build (round-2 master path)" >> $W
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_alt.so timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $W 2>&1
echo "== prof build" >> $W
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so timeout 600 python scripts/quick_bench.py suzanne,64,64,512,0 ce,32,32,1024,0 >> $W 2>&1
grep -v amdgpu.ids $W
