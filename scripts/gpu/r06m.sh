#!/bin/bash
# round 6 (second session): where the generator wave sits.  K: GEN_ACC + one note per round (generator = hardware wave 4,
# beside the frontier wave); N: generator = hardware wave 0 (beside tracing role 3); P: N + s_setprio 3 in the tracing
# waves; Q: K + s_setprio 3 in the tracing waves.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06m; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in K N P Q; do
    lib=$L/libptw_hip_pw$v.so
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib timeout 120 python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
grep -A1 "variant" $out/ab.txt | grep -v "^--"
for v in P; do
  PTW_LIB_PATH=$L/libptw_hip_pw$v.so timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels" > $out/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $out/parity_$v.log)"
done
