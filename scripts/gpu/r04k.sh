#!/bin/bash
# round 4: s_memtime anatomy of the worker-wave kernels AS SHIPPED (prof build), every worker wave's
# busy time printed: which waves set the tick?
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04k; mkdir -p $O
L=$PWD/pt-three-ways_amd
prof() { echo "== $1 $2x$3 @ $4 rows 0:$5" | tee -a $O/anatomy.txt
  PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 300 python bench.py --scene $1 --width $2 --height $3 --spp $4 --rows 0:$5 --no-cpu-baseline --no-parity --no-secondary 2>&1 \
    | grep -v amdgpu.ids | grep "PHASES\|MASTER\|WORKER" | tee -a $O/anatomy.txt; }
prof suzanne 1024 1024 512 8
prof ce 2048 2048 1024 1
prof suzanne 1024 1024 256 8
