#!/bin/bash
# round 6: prefilter (both forms) parity + timing, the new one-master instantiation tests, the dispatch sweep
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06e; mkdir -p $out
python -m pytest tests/test_gpu_accel.py tests/test_gpu_round6.py -x -q -m gpu -k "accel or prefilter or obj_scene or one_master" > $out/parity.log 2>&1
tail -6 $out/parity.log
python scripts/quick_bench.py ce,512,512,16,1 ce,512,512,16,1,accel=2 ce,512,512,16,1,accel=2,pix_kernel=1 \
      suzanne,1024,1024,16,1 suzanne,1024,1024,16,1,accel=2 suzanne,1024,1024,16,1,accel=2,pix_kernel=1 \
      bbc-owl,512,512,32,1 bbc-owl,512,512,32,1,accel=2 cornell,512,512,32,1 cornell,512,512,32,1,accel=2 > $out/prefilter_ab.txt 2>&1
cat $out/prefilter_ab.txt
python scripts/dispatch_sweep.py $out/dispatch_sweep.md > $out/dispatch_sweep.log 2>&1
tail -12 $out/dispatch_sweep.log
