#!/bin/bash
# round 6 (second session): GEN_ACC + the guess histogram kept by the generator wave too (PTW_SPEC_GEN_HIST):
#   base  the tree's library     K  GEN_ACC + one histogram note per round (tracing waves)     L  GEN_ACC + GEN_HIST
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06l; mkdir -p $out
L=$PWD/pt-three-ways_amd
for v in L; do
  PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $out/parity_$v.log)"
done
for rep in 1 2 3; do
  for v in base K L; do
    if [ $v = base ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
grep -A1 "variant" $out/ab.txt | grep -v "^--"
