#!/bin/bash
# Round 3, GPU call 5: worker-wave kernels - master priority x balance ratio.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
W=$OUT/prio_ab.txt
: > $W
run() { local label=$1; shift; echo "== $label" >> $W; ( env "$@" timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $W 2>&1 ); }
run "prio 0 ratio 100" PTW_SEQ_PRIO=0 PTW_SEQ_BALANCE=100
run "prio 3 ratio 100" PTW_SEQ_PRIO=3 PTW_SEQ_BALANCE=100
run "prio 3 ratio 60" PTW_SEQ_PRIO=3 PTW_SEQ_BALANCE=60
run "prio 3 ratio 30" PTW_SEQ_PRIO=3 PTW_SEQ_BALANCE=30
run "prio 0 ratio 30" PTW_SEQ_PRIO=0 PTW_SEQ_BALANCE=30
run "prio 1 ratio 60" PTW_SEQ_PRIO=1 PTW_SEQ_BALANCE=60
grep -v amdgpu.ids $W
