#!/bin/bash
# round 6 (second session): the GENERATOR wave adds the committed radiance and stores the sample (PTW_SPEC_GEN_ACC):
#   base  the tree's library        F  wave 1 accumulates + one histogram note per round (the best of r06j)
#   J     GEN_ACC                   K  GEN_ACC + one histogram note per round
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06k; mkdir -p $out
L=$PWD/pt-three-ways_amd
# parity first: the byte-equality tests between the sequential kernels, goldens, the whole headline frame
for v in K; do
  PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $out/parity_$v.log)"
done
for rep in 1 2 3; do
  for v in base F J K; do
    if [ $v = base ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
