#!/bin/bash
# round 5: why does the two-rank bench line say rccl_log null?  Look at RCCL's debug file on the box.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05k}; mkdir -p $O
env | grep -i "nccl\|rccl" ; ( PTW_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --width 128 --height 128 --spp 32 --no-cpu-baseline --no-parity --no-strict --no-other-configs > $O/two.json 2> $O/two.err; echo "rc=$?" )
python - <<PY
import json
line=[l for l in open("$O/two.json") if l.startswith("{")][-1]
print(json.loads(line).get("rccl_transport"))
PY
ls -la /tmp/ptw_bench_rccl* 2>&1 | head
for f in /tmp/ptw_bench_rccl*; do echo "== $f"; grep -c " via " $f; grep -i "via\|channel\|transport" $f | head -12; echo "-- head"; head -15 $f; done 2>&1 | cut -c1-220 > $O/rccl_logs.txt
head -80 $O/rccl_logs.txt
