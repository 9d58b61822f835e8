#!/bin/bash
# Round 3, GPU call 1: new parity tests + whole GPU suite, EXEC-mask microbenchmark, PERPIXEL A/B at
# the BASELINE shape with clock / power sampling.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showclocks --showpower --showperflevel > $OUT/smi_start.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_round3.py -q -m gpu -x --durations=15 > $OUT/pytest_round3.log 2>&1; echo "rc=$?" >> $OUT/pytest_round3.log )
tail -5 $OUT/pytest_round3.log
( timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round3.py > $OUT/pytest_rest.log 2>&1; echo "rc=$?" >> $OUT/pytest_rest.log )
tail -5 $OUT/pytest_rest.log
timeout 120 scripts/microbench/exec_mask_cost > $OUT/exec_mask_cost.txt 2>&1
cat $OUT/exec_mask_cost.txt

# ---- PERPIXEL A/B, cornell 1024 x 1024 @ 256 spp, every variant twice, clocks sampled meanwhile ----
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk"; sleep 0.25; done ) > $OUT/smi_samples.txt 2>&1 &
SMI=$!
AB=$OUT/perpixel_ab.txt
: > $AB
run() { # label, env..., -- args
  local label=$1; shift
  echo "== $label  [$(date +%s.%N)]" >> $AB
  ( env "$@" timeout 300 python scripts/quick_bench.py cornell,1024,1024,256,1 cornell,1024,1024,256,1 >> $AB 2>&1 )
}
run "persistent (default)" PTW_PIX_KERNEL=persistent
run "lockstep spl=1" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=1
run "lockstep spl=4" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=4
run "lockstep spl=16" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=16
run "lockstep spl=64" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=64
run "lockstep spl=256" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=256
run "persistent, fused uv test (alt lib)" PTW_PIX_KERNEL=persistent PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_alt.so
run "lockstep spl=1, fused uv test (alt lib)" PTW_PIX_KERNEL=legacy PTW_LIB_PATH=$REPO/pt-three-ways_amd/libptw_hip_alt.so
run "persistent again" PTW_PIX_KERNEL=persistent
run "lockstep spl=1 again" PTW_PIX_KERNEL=legacy PTW_PIX_SPL=1
kill $SMI
# ---- the u-first early-out on the scenes it is for ----
UF=$OUT/ufirst_ab.txt
: > $UF
for lib in libptw_hip.so libptw_hip_alt.so; do
  for k in persistent legacy; do
    echo "== $lib $k" >> $UF
    PTW_LIB_PATH=$REPO/pt-three-ways_amd/$lib PTW_PIX_KERNEL=$k timeout 300 python scripts/quick_bench.py suzanne,1024,1024,32,1 ce,512,512,16,1 bbc-owl,512,512,64,1 >> $UF 2>&1
  done
done
cat $AB $UF
rocm-smi --showclocks --showpower > $OUT/smi_end.txt 2>&1
