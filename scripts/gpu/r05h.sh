#!/bin/bash
# round 5: the camera in LDS (kernel-argument diet) and the worker-side pick of the two-master kernels
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05h}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_cli.py tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu \
  -k "two_master or ties or dropped or paired or cfg1 or speculative or golden" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $O/pytest_subset.log
timeout 400 python scripts/quick_bench.py cornell,1024,64,256,0 cornell,1024,64,256,0 cornell,1024,1024,256,0 \
    suzanne,256,256,512,0,seq_worker_pick=1 suzanne,256,256,512,0,seq_worker_pick=0 suzanne,256,256,512,0,seq_worker_pick=1 \
    suzanne,256,256,256,0 ce,2048,8,1024,0,seq_worker_pick=0 ce,2048,8,1024,0,seq_worker_pick=1 2>&1 | grep -v amdgpu.ids > $O/ab.txt; cat $O/ab.txt
