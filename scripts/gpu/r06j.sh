#!/bin/bash
# round 6 (second session), second look at the commit phase of the headline kernel (scripts/mkvar_spec.sh variants):
#   base  the tree's library
#   D     PTW_SPEC_ACC_WAVE=1 (wave 1 adds the committed radiance and stores the sample)
#   F     D + only the frontier's own count feeds the guess histogram (one `note` per round instead of five)
#   G     F + rays counted and the generator command written by wave 1
#   H     G + the guesses refreshed every 8th pixel (the histogram still halved every pixel)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06j; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in base D F G H; do
    if [ $v = base ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib python scripts/quick_bench.py cornell,512,512,256,0 example1,256,256,256,0 single-sphere,256,256,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
for v in H; do
  PTW_LIB_PATH=$L/libptw_hip_pw$v.so python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $out/parity_$v.log)"
done
