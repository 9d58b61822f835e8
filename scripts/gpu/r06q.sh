#!/bin/bash
# round 6 (second session): fewer scalar registers live across a round (S1: the tracing waves' ray counter is a
# constant - one primary ray per pixel -, the cross-pixel decision is taken in the pixel's last round instead of from
# six values carried through every round) against variant K.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06q; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in S1 K; do
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
PTW_LIB_PATH=$L/libptw_hip_pwS1.so timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity.log 2>&1
echo "parity: $(tail -1 $out/parity.log)"
