#!/bin/bash
# Round 3, GPU call 4: traceSequentialGang with lean synchronisation; finer phases of the worker-wave master.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
( timeout 600 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "gang" > $OUT/pytest_gang.log 2>&1; echo "rc=$?" >> $OUT/pytest_gang.log )
tail -4 $OUT/pytest_gang.log
AB=$OUT/gang_ab.txt
: > $AB
runargs() { local label=$1; local args=$2; shift; shift; echo "== $label" >> $AB; ( env "$@" timeout 300 python scripts/quick_bench.py $args >> $AB 2>&1 ); }
runargs "32 passes, one CU per pass (traceSequentialSpec)" "cornell,512,512,32,0" PTW_SEQ_GANG=0
runargs "32 passes, 8 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=8
runargs "32 passes, 4 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=4
runargs "32 passes, 2 CUs per pass" "cornell,512,512,32,0" PTW_SEQ_GANG=2
runargs "64 passes, 4 CUs per pass" "cornell,512,512,64,0" PTW_SEQ_GANG=4
runargs "single-sphere 32 passes: 8 CUs" "single-sphere,512,512,32,0" PTW_SEQ_GANG=8
grep -v amdgpu.ids $AB
for g in 8 4; do
  echo "== prof build, $g CUs per pass" >> $OUT/gang_phases.txt
  PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so PTW_SEQ_GANG=$g timeout 300 python scripts/quick_bench.py cornell,256,256,32,0 >> $OUT/gang_phases.txt 2>&1
done
grep -E "^==|cornell|member 0 wave 0|member 7 wave 3|member 3 wave 1" $OUT/gang_phases.txt | tail -24
echo "== prof build: phases, suzanne two masters / one master, ce" > $OUT/worker_phases.txt
PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_prof.so timeout 600 python scripts/quick_bench.py suzanne,64,64,512,0 suzanne,64,64,256,0 ce,32,32,1024,0 >> $OUT/worker_phases.txt 2>&1
grep -v amdgpu.ids $OUT/worker_phases.txt
