#!/bin/bash
# round 6: the headline kernel with the NEXT pixel's camera ray + first hit traced ahead by the waves that have no
# sub-sample left in a pixel's last round (traceSequentialSpec, CROSS) against round 5's form of the same kernel
# (--debug seq_small_kernel=3): parity of both first, then same-box A/B at three frame sizes, alternating.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06c; mkdir -p $out
python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu \
  -k "sequential_kernel_variants or small_scene_kernels or headline or golden or parity or full" > $out/parity.log 2>&1
tail -5 $out/parity.log
for rep in 1 2; do
  python scripts/quick_bench.py cornell,256,256,256,0 cornell,256,256,256,0,seq_small_kernel=3 \
      cornell,512,512,256,0 cornell,512,512,256,0,seq_small_kernel=3 \
      single-sphere,256,256,256,0 single-sphere,256,256,256,0,seq_small_kernel=3 \
      example1,256,256,256,0 example1,256,256,256,0,seq_small_kernel=3
done > $out/ab.txt 2>&1
cat $out/ab.txt
