#!/bin/bash
# round 6: PTW_ACCEL_PREFILTER (the conservative fp32 look before the fp64 triangle test) - parity, then same-box
# timing against the brute-force PERPIXEL kernels and the BVH mode; and the commit histogram of the headline
# kernel with the cross-pixel candidate (instrumented build).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06d; mkdir -p $out
python -m pytest tests/test_gpu_accel.py tests/test_gpu_parity.py -x -q -m gpu -k "accel or prefilter or intersect or kat" > $out/parity.log 2>&1
tail -8 $out/parity.log
for rep in 1 2; do
  python scripts/quick_bench.py ce,512,512,16,1 ce,512,512,16,1,pix_kernel=1 ce,512,512,16,1,accel=2 ce,512,512,16,1,accel=1 \
      suzanne,1024,1024,16,1 suzanne,1024,1024,16,1,pix_kernel=1 suzanne,1024,1024,16,1,accel=2 suzanne,1024,1024,16,1,accel=1 \
      cornell,1024,1024,32,1,pix_kernel=1 cornell,1024,1024,32,1,accel=2 \
      bbc-owl,512,512,32,1 bbc-owl,512,512,32,1,accel=2
done > $out/prefilter_ab.txt 2>&1
cat $out/prefilter_ab.txt
PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so python scripts/quick_bench.py cornell,64,64,256,0 cornell,64,64,256,0,seq_small_kernel=3 > $out/spec_histogram.txt 2>&1
grep -v "^SPEC wave [123]" $out/spec_histogram.txt | head -20
