#!/bin/bash
# round 6 (second session): the first-bounce tables (FAN: four generator waves, csrc/seq_spec.hip): parity, then
# the tree's library (FAN) against the same source without the tables (NF: -DPTW_SPEC_FAN=0) and variant K.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06p; mkdir -p $out
L=$PWD/pt-three-ways_amd
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity.log 2>&1
echo "parity: $(tail -1 $out/parity.log)"
for rep in 1 2 3; do
  for v in tree NF K; do
    if [ $v = tree ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
