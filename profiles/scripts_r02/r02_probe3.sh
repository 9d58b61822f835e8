# wide v3 (level-granular scheduler) vs the four-wave kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
timeout 300 python -m pytest tests/test_gpu_cli.py -m gpu -q -x -k "variants" 2>&1 | tail -4
for env in "PTW_SEQ_WIDE=1 PTW_WIDE_G=8" "PTW_SEQ_WIDE=1 PTW_WIDE_G=16" "PTW_SEQ_WIDE=0"; do
  echo "== $env"; env $env timeout 120 python scripts/quick_bench.py cornell,256,256,256,0 2>&1 | tail -1
done
echo "== phases G=8"; PTW_SEQ_WIDE=1 PTW_LIB_PATH=$L/libptw_hip_prof.so PTW_WIDE_G=8 timeout 120 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "WIDE|Msamples" | head -10
echo "== phases G=16"; PTW_SEQ_WIDE=1 PTW_LIB_PATH=$L/libptw_hip_prof.so PTW_WIDE_G=16 timeout 120 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "WIDE|Msamples" | head -10
