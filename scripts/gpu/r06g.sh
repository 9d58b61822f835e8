#!/bin/bash
# round 6: PTW_ACCEL_PREFILTER under the SEQUENTIAL policy (worker lanes look in fp32 first): parity of every
# instantiation, then same-box timing against the plain worker-wave kernels on cfg3 / cfg4 sub-runs.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06g; mkdir -p $out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_accel.py -x -q -m gpu -k "sequential_prefilter or accel_mode_matches" --durations=5 > $out/parity.log 2>&1
tail -8 $out/parity.log
for rep in 1 2; do
python scripts/quick_bench.py ce,2048,8,1024,0 ce,2048,8,1024,0,accel=2 suzanne,1024,128,512,0 suzanne,1024,128,512,0,accel=2 \
   suzanne,512,256,256,0 suzanne,512,256,256,0,accel=2 ce,512,32,256,0 ce,512,32,256,0,accel=2
done > $out/seq_prefilter_ab.txt 2>&1
cat $out/seq_prefilter_ab.txt
