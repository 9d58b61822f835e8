#!/bin/bash
# round 6: prefilter after the instruction diet (both forms) - parity + timing; the small-scene calibration;
# the first-contact kit's dry run; the dispatch sweep again (does the table hold after the fix?)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06f; mkdir -p $out
python -m pytest tests/test_gpu_accel.py tests/test_gpu_round6.py -x -q -m gpu -k "accel or prefilter or obj_scene or one_master or small_scene or first_contact" --durations=6 > $out/parity.log 2>&1
tail -14 $out/parity.log
for rep in 1 2; do
python scripts/quick_bench.py ce,512,512,16,1 ce,512,512,16,1,accel=2 suzanne,1024,1024,16,1 suzanne,1024,1024,16,1,accel=2 \
      cornell,512,512,32,1 cornell,512,512,32,1,accel=2 cornell,512,512,32,1,pix_kernel=1 cornell,512,512,32,1,accel=2,pix_kernel=1
done > $out/prefilter_ab.txt 2>&1
cat $out/prefilter_ab.txt
