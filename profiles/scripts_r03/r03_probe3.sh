#!/bin/bash
# Round 3, GPU call 3: traceSequentialGang (several CUs per pass) - parity, then the measurement the
# verdict asked for: cfg2's per-GPU share at 8 GPUs (32 passes) on ONE GPU, against the one-CU kernel.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
( timeout 600 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "gang" > $OUT/pytest_gang.log 2>&1; echo "rc=$?" >> $OUT/pytest_gang.log )
tail -15 $OUT/pytest_gang.log
( timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "sequential_kernel_variants" > $OUT/pytest_cli.log 2>&1; echo "rc=$?" >> $OUT/pytest_cli.log )
tail -5 $OUT/pytest_cli.log
AB=$OUT/gang_ab.txt
: > $AB
run() { local label=$1; shift; echo "== $label" >> $AB; ( env "$@" timeout 300 python scripts/quick_bench.py >> $AB 2>&1 ); }
runargs() { local label=$1; local args=$2; shift; shift; echo "== $label" >> $AB; ( env "$@" timeout 300 python scripts/quick_bench.py $args >> $AB 2>&1 ); }
runargs "32 passes, one CU per pass (traceSequentialSpec)" "cornell,512,512,32,0" PTW_SEQ_GANG=0
runargs "32 passes, 8 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=8
runargs "32 passes, 4 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=4
runargs "32 passes, 2 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=2
runargs "64 passes, one CU per pass" "cornell,512,512,64,0" PTW_SEQ_GANG=0
runargs "64 passes, 4 CUs per pass" "cornell,512,512,64,0" PTW_SEQ_GANG=4
runargs "128 passes, one CU per pass" "cornell,512,512,128,0" PTW_SEQ_GANG=0
runargs "128 passes, 2 CUs per pass" "cornell,512,512,128,0" PTW_SEQ_GANG=2
runargs "single-sphere 32 passes: one CU / 8 CUs" "single-sphere,512,512,32,0" PTW_SEQ_GANG=0
runargs "single-sphere 32 passes: 8 CUs" "single-sphere,512,512,32,0" PTW_SEQ_GANG=8
runargs "example1 32 passes: one CU" "example1,512,512,32,0" PTW_SEQ_GANG=0
runargs "example1 32 passes: 8 CUs" "example1,512,512,32,0" PTW_SEQ_GANG=8
cat $AB
# phase counters of the debug build
if [ -f pt-three-ways_amd/libptw_hip_prof.so ]; then
  for g in 8 4; do
    echo "== prof build, $g CUs per pass" >> $OUT/gang_phases.txt
    PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so PTW_SEQ_GANG=$g timeout 300 python scripts/quick_bench.py cornell,256,256,32,0 >> $OUT/gang_phases.txt 2>&1
  done
  echo "== prof build, one CU per pass" >> $OUT/gang_phases.txt
  PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so PTW_SEQ_GANG=0 timeout 300 python scripts/quick_bench.py cornell,256,256,32,0 >> $OUT/gang_phases.txt 2>&1
  cat $OUT/gang_phases.txt
fi
# ---- worker-wave kernels after the unrolling fix: ce must be back at >= 2.0; phase counters of the new master path ----
W=$OUT/worker_ab.txt
: > $W
echo "== shipped build" >> $W
timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $W 2>&1
echo "== alt build (round-2 master path)" >> $W
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_alt.so timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $W 2>&1
for r in 150 200; do
  echo "== shipped build, PTW_SEQ_BALANCE=$r" >> $W
  PTW_SEQ_BALANCE=$r timeout 600 python scripts/quick_bench.py ce,256,128,1024,0 >> $W 2>&1
done
echo "== prof build: phases, suzanne two masters / one master, ce" >> $W
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so timeout 600 python scripts/quick_bench.py suzanne,64,64,512,0 suzanne,64,64,256,0 ce,32,32,1024,0 >> $W 2>&1
cat $W
