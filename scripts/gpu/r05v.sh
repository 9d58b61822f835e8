#!/bin/bash
# round 5: the headline frame in ONE band launch (staging budget 7 GB: the frame stages 6.4 GB) against the
# default two (4 GiB) - what the join of 256 unequal chains at a band's end costs
export TMPDIR=/tmp
O=gpurun_out/${1:-r05v}; mkdir -p $O
( timeout 100 python scripts/quick_bench.py cornell,1024,1024,256,0 2>&1 | grep -v amdgpu.ids ) > $O/two_bands.txt &
wait
( PTW_STAGE_BUDGET_MB=7000 timeout 100 python scripts/quick_bench.py cornell,1024,1024,256,0 2>&1 | grep -v amdgpu.ids ) > $O/one_band.txt
cat $O/two_bands.txt $O/one_band.txt
