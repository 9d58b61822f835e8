#!/bin/bash
# round 5, first GPU call: the paired two-master kernels (two sub-samples in flight per master) and the
# ABI-v5 plumbing - parity of everything two-master first, then a quick paired / single-ray A/B on the
# BASELINE scenes, the whole GPU suite, and the instrumented build's per-request anatomy.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_cli.py -x -q -m gpu \
  -k "two_master or ties or dropped" > $O/pytest_two_master.log 2>&1; echo "two-master pytest rc=$?"; tail -5 $O/pytest_two_master.log
timeout 300 python scripts/quick_bench.py suzanne,1024,128,512,0,seq_pairing=1 suzanne,1024,128,512,0,seq_pairing=0 \
  suzanne,1024,128,512,0,seq_pairing=1 ce,2048,8,1024,0,seq_pairing=1 ce,2048,8,1024,0,seq_pairing=0 \
  ce,2048,8,1024,0,seq_pairing=1,seq_units=9:9:9 > $O/ab.txt 2>&1; cat $O/ab.txt | grep -v amdgpu.ids
PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so timeout 300 python scripts/quick_bench.py \
  suzanne,64,64,512,0,seq_pairing=1 suzanne,64,64,512,0,seq_pairing=0 ce,32,32,1024,0,seq_pairing=1 ce,32,32,1024,0,seq_pairing=0 \
  > $O/anatomy.txt 2>&1; grep -v amdgpu.ids $O/anatomy.txt | cut -c1-400
( timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log ); tail -6 $O/pytest_gpu.log
