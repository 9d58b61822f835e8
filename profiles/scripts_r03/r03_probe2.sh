#!/bin/bash
# Round 3, GPU call 2: the worker-wave kernels' master fast path (6-way uniform pick, chainMaster,
# deferred stack entry) and the balance ratio, on suzanne and ce; parity tests of the same build.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log )
tail -4 $OUT/pytest.log
AB=$OUT/seq_ab.txt
: > $AB
run() { local label=$1; shift; echo "== $label" >> $AB; ( env "$@" timeout 600 python scripts/quick_bench.py suzanne,512,512,512,0 ce,256,128,1024,0 suzanne,512,512,256,0 >> $AB 2>&1 ); }
run "round-2 master path (alt lib: general pick, radianceChain), ratio 100" PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_alt.so
run "pick6 + chainMaster, ratio 100" PTW_SEQ_BALANCE=100
run "pick6 + chainMaster, ratio 150" PTW_SEQ_BALANCE=150
run "pick6 + chainMaster, ratio 200" PTW_SEQ_BALANCE=200
run "pick6 + chainMaster, ratio 250" PTW_SEQ_BALANCE=250
run "pick6 + chainMaster, ratio 300" PTW_SEQ_BALANCE=300
run "pick6 + chainMaster, ratio 60" PTW_SEQ_BALANCE=60
cat $AB
