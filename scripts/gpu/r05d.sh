#!/bin/bash
# round 5, fourth GPU call: interleaved protocol - one barrier per tick, both masters busy, slots alternate.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_cli.py -x -q -m gpu \
  -k "two_master or ties or dropped" > $O/pytest_two_master.log 2>&1; echo "two-master pytest rc=$?"; tail -3 $O/pytest_two_master.log
timeout 300 python scripts/quick_bench.py suzanne,1024,128,512,0,seq_pairing=1 suzanne,1024,128,512,0,seq_pairing=0 \
  suzanne,256,256,512,0,seq_pairing=1 suzanne,256,256,512,0,seq_pairing=0 \
  ce,2048,8,1024,0,seq_pairing=1 ce,2048,8,1024,0,seq_pairing=0 cornell,1024,64,256,0 cornell,1024,64,256,0 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so timeout 300 python scripts/quick_bench.py \
  suzanne,64,64,512,0,seq_pairing=1 ce,32,32,1024,0,seq_pairing=1 > $O/anatomy.txt 2>&1; grep -v amdgpu.ids $O/anatomy.txt | cut -c1-420
