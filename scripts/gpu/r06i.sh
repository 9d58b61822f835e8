#!/bin/bash
# round 6 (second session): housekeeping of the headline kernel's commit phase moved off the frontier wave.
# Variants of csrc/seq_spec.hip built by scripts/mkvar_spec.sh (libptw_hip_pw<X>.so), same box, alternating:
#   base  the tree's library (wave 0 adds the committed radiance, stores the sample, counts rays, writes the generator command)
#   A     PTW_SPEC_ACC_WAVE=3 (the wave with the most slack adds the radiance and stores the sample)
#   D     PTW_SPEC_ACC_WAVE=1
#   E     A + rays counted and the generator command written by that wave too
#   C     only the frontier's own count feeds the guess histogram, guesses refreshed every 8th pixel
#   B     E + C
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06i; mkdir -p $out
L=$PWD/pt-three-ways_amd
for rep in 1 2 3; do
  for v in base A D E C B; do
    if [ $v = base ]; then lib=$L/libptw_hip.so; else lib=$L/libptw_hip_pw$v.so; fi
    echo "== variant $v (rep $rep)"
    PTW_LIB_PATH=$lib python scripts/quick_bench.py cornell,512,512,256,0 2>&1 | grep Msamples
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
# parity of the most aggressive variant (B) and of A: the sequential-kernel byte-equality and golden tests
for v in B A; do
  PTW_LIB_PATH=$L/libptw_hip_pw$v.so python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round6.py -x -q -m gpu \
    -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $out/parity_$v.log)"
done
# the ce WRITE_SIZE question (profiles/r06z_pmc_kernels.txt: 59 B/sample where round 5 measured 24.6): twice
for rep in 1 2; do PMC="WRITE_SIZE" bash scripts/pmc_quick.sh ce,64,64,1024,0 2>&1 | grep -v amdgpu.ids | grep "Msamples\|{" | tail -3; done > $out/ce_write_size_again.txt 2>&1
cat $out/ce_write_size_again.txt
